set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; tail -6 gpurun_out/gpu_tests.log
timeout -k 10 600 python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err; tail -c 600 gpurun_out/bench_r03.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python3 bench.py --steps 20 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
export MG_FUSED=2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 tools/kernel_probe.py 4097 10 jacobi sweeps2 down_leg up_leg span_leg span_leg_nomid > gpurun_out/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 tools/kernel_probe.py 4097 10 jacobi sweeps2 down_leg up_leg span_leg span_leg_nomid > gpurun_out/pmc_w.log 2>&1
unset MG_FUSED
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_w -o w -- python3 tools/wcycle_probe.py > gpurun_out/prof_w.log 2>&1
timeout -k 10 400 python tools/config_times.py all > gpurun_out/config_times.log 2>&1; cat gpurun_out/config_times.log
find gpurun_out -name "*.csv" | head -30
