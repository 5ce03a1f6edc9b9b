timeout -k 10 200 python3 tools/exp_libs.py --workload blobs100m --reps 5 --kernels exp/lib_even.so 2>&1 | grep -v "l[1-7]\|amdgpu"
PC_EXP_EVEN=1 timeout -k 10 200 python3 tools/exp_libs.py --workload blobs100m --reps 5 --kernels exp/lib_even.so 2>&1 | grep -v "l[1-7]\|amdgpu"
