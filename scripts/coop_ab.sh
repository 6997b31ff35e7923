set -e
for v in coop nocoop; do if [ $v = nocoop ]; then export SCRUBBY_HIP_NO_COOP=1; else unset SCRUBBY_HIP_NO_COOP; fi
SCRUBBY_HIP_DBG=16 timeout -k 10 150 python bench.py --workload ont --steps 1 --warmup 1 --no-cpu > gpurun_out/coop_$v.log 2>&1 || { echo "FAILED $v"; tail -5 gpurun_out/coop_$v.log; exit 1; }
grep "^{" gpurun_out/coop_$v.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['stage_ms_per_step']['extension stage (k_long_chains + k_regs_align_long)'], d['result']['reads_removed'], d['result']['rmq_exact'], d['result']['ext_unresolved'])"
grep "part 2\|exact long join" gpurun_out/coop_$v.log | tail -4 | cut -c1-200
done
