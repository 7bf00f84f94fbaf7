# The two rocprofv3 PMC passes of the level-0 kernels, their summary (profiles/pmc_latest.json, stamped with the library's source
# hash) and the bench line that reads it, in ONE run on one box: roofline.traffic_is_from_this_build is then true.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MG_FUSED=2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 tools/kernel_probe.py 4097 10 jacobi sweeps2 down_leg up_leg span_leg span_leg_nomid > gpurun_out/pmc_f.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 tools/kernel_probe.py 4097 10 jacobi sweeps2 down_leg up_leg span_leg span_leg_nomid > gpurun_out/pmc_w.log 2>&1
unset MG_FUSED
python3 tools/pmc_summary.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv profiles/pmc_latest.json | grep span
timeout -k 10 600 python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err
cp profiles/pmc_latest.json gpurun_out/pmc_latest.json
python -c "
import json;d=json.loads(open('gpurun_out/bench_r03.json').read().strip().splitlines()[-1]);r=d['roofline'];print(d['value'],d['ms_per_step'],r['frac'],r['traffic'],r['traffic_is_from_this_build'],r['launch_ms'])"
