import os, sys
sys.argv=[sys.argv[0]]
exec(open('scripts/gpu_vqe_timing.py').read().split("run(16, 4, 4)")[0])
run(28, 12, 2, reps=2)
run(28, 12, 8, reps=1)
