#!/usr/bin/env python3
"""Per-step view of a rocprofv3 kernel trace of the headline run (tools/gpu_job.sh timeline): every noise kernel's duration and the gap to the next one's start, and the kernel
sequence of ONE pipeline between two of its noise kernels (its erosion, kernel by kernel, with start offsets): where a map's latency goes beside the other maps' noise.
usage: timeline_steps.py <trace dir> [steps=20]"""
import csv, glob, sys
D = sys.argv[1]; K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows=[]
for f in glob.glob(D + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id') or r.get('Stream_Id')))
rows.sort()
noise=[r for r in rows if 'k_sine_grid' in r[2]]
# last 20 noise kernels = timed region
noise=noise[-K:]
t0=noise[0][0]
def short(n):
    if 'k_sine_grid' in n: return 'SINE'
    if 'k_waves_lean' in n: return 'lean_trace'
    if 'k_waves_nolds' in n: return 'check/commit'
    if 'gen_grid_dev' in n: return 'tables'
    if 'sparse_erosion' in n: return 'sp_'+n.split('EUlmE')[-1][:6]
    if 'apply_erosion_dev' in n: return 'ero_'+n.split('EUlmE')[-1][:6]
    if 'copyBuffer' in n: return 'copy'
    if 'fillBuffer' in n: return 'fill'
    return n[:30]
for i in range(1,len(noise)):
    a=noise[i-1]; b=noise[i]
    gap=(b[0]-a[1])/1e3
    print(f"noise {i-1} q{a[3]} dur {(a[1]-a[0])/1e3:7.1f} us, gap to next start {gap:7.1f} us")
# one erosion sequence in detail: pick queue of noise[5], list kernels on that queue between noise[5] end and its next noise start
q=noise[5][3]
seq=[r for r in rows if r[3]==q and r[0]>=noise[5][0]]
nxt=[r for r in seq if 'k_sine_grid' in r[2] and r[0]>noise[5][0]]
end=nxt[0][0] if nxt else seq[-1][1]
print('--- queue',q,'from its noise start to its next noise start: ',(end-noise[5][0])/1e3,'us')
for r in seq:
    if r[0]>end: break
    print(f"  {short(r[2]):14s} start {(r[0]-noise[5][0])/1e3:8.1f} dur {(r[1]-r[0])/1e3:7.1f}")
