# the driver's own command (default flags: headline + the `secondary` object), then the kernel timelines of one step of each workload
timeout -k 10 900 python bench.py > gpurun_out/r05_driver_like.log 2>&1 || { tail -5 gpurun_out/r05_driver_like.log; exit 1; }
grep "^{" gpurun_out/r05_driver_like.log | tail -1 > gpurun_out/r05_driver_like.json
python3 -c "
import json; d=json.load(open('gpurun_out/r05_driver_like.json')); print(d['value'], d['ms_per_step'], {k:(v.get('value'), v.get('ms_per_step')) for k,v in d['secondary'].items() if isinstance(v,dict)})"
bash scripts/step_timeline.sh 20m && bash scripts/step_timeline.sh ont --workload ont --records 1000000 --steps 1
