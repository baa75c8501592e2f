# rocprofv3 kernel trace of the exact path on C1 and C3 (tools/profile_exact.py) + the direct-solver parity tests
export TMPDIR=/tmp
for c in c1 c3; do rm -rf gpurun_out/prof_exact_$c; timeout 200 rocprofv3 --kernel-trace -d gpurun_out/prof_exact_$c -o exact -- python tools/profile_exact.py $c 5 > gpurun_out/exact_$c.log 2>&1; grep "run [14]" gpurun_out/exact_$c.log; done
timeout 300 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
