mkdir -p gpurun_out/r06c; O=gpurun_out/r06c
python -m pytest tests/test_gpu_fused.py -q -s 2>&1 | grep -v "^fused grid" | tail -25 > $O/fused.log
for f in 0 1; do
  TERRA_GEN_FUSED=$f python tools/prof_voxels.py 512 512 >> $O/times.txt 2>&1
  TERRA_GEN_FUSED=$f python tools/prof_voxels.py 512 64 >> $O/times.txt 2>&1
  TERRA_GEN_FUSED=$f python tools/prof_noise.py 16384 10 0 8 2>&1 | head -1 >> $O/times.txt
  TERRA_GEN_FUSED=$f python tools/prof_noise.py 4096 10 0 8 2>&1 | head -1 >> $O/times.txt
done
cat $O/fused.log $O/times.txt
