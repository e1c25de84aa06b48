pr() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), d.get('value_reference_schedule'))"; }
cd $GRAFT_REPO_ROOT
PRESIGHT_BENCH_LIVE_TABLES=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --psnr-steps 0 --no-secondary 2>/dev/null | tail -1 | pr cur_20_5
PRESIGHT_BENCH_LIVE_TABLES=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --psnr-steps 0 --no-secondary 2>/dev/null | tail -1 | pr cur_10_3
cd _wt_r05
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --psnr-steps 0 --no-secondary 2>/dev/null | tail -1 | pr r05_20_5
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --psnr-steps 0 --no-secondary 2>/dev/null | tail -1 | pr r05_10_3
