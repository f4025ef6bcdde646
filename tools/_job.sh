mkdir -p gpurun_out/r06d; O=gpurun_out/r06d
python -m pytest tests/test_gpu_fused.py -q -s -x 2>&1 | grep -v "^fused grid" | tail -25 > $O/fused.log
for f in 0 1; do
  TERRA_GEN_FUSED=$f python tools/prof_noise.py 4096 10 1,2,4 8 2>&1 | head -3 >> $O/times.txt
  TERRA_GEN_FUSED=$f python tools/prof_noise.py 16384 3 1,2,4 8 2>&1 | head -3 >> $O/times.txt
  TERRA_GEN_FUSED=$f python tools/prof_voxels.py 512 64 1,2 >> $O/times.txt 2>&1
done
cat $O/fused.log $O/times.txt
