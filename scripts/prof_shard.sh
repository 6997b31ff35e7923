cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3shard -- python3 bench.py --records 2500000 --steps 2 --warmup 1 --no-cpu > gpurun_out/r3shard.log 2>&1 || exit 1
