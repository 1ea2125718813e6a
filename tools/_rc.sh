run() { python bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 30 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d.get('parity',{}).get('rel_l2_out'), d['roofline_block_fwd']['us_per_launch'])"; }
PIT_LIB_PATH=$PWD/_diag/libpit_vb64.so run "E as fragments"
run "E through LDS"
PIT_LIB_PATH=$PWD/_diag/libpit_vb64.so run "E as fragments"
run "E through LDS"
python -m pytest tests -m gpu -q -x -k "processor or block" 2>&1 | tail -2
