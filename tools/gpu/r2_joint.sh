R=$GRAFT_REPO_ROOT
cd $R
b() { python bench.py --game all16 --num-envs 16384 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; }
echo "default"; b
echo "queues 16"; GPU_MAX_HW_QUEUES=16 b
echo "threads 1"; PROCGEN_AMD_HOST_THREADS=1 b
python -m pytest tests -m gpu -x -q -k "joint or sharded or sixteen" 2>&1 | tail -2
