cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -k "count_tiles or liop_match or split_mfma or test_stage" 2>&1 | tail -4
timeout 900 python bench.py --config liop144c --steps 2 --warmup 1 --no-cpu-baseline --no-stage-leg 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); o=j.get('opt_in_split_mfma',{}); print(json.dumps({'value':j['value'],'opt_in':{k:o.get(k) for k in ('value','ms_per_step','identical_to_headline_graphs','roofline','exact_fallback_queries')}})[:1500])"
timeout 600 python bench.py --config stage --steps 3 --warmup 1 --images 24 --stage-quick 2>/dev/null | tail -1 | cut -c1-900
