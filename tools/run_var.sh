O=gpurun_out/var; mkdir -p $O
run() { tag=$1; shift; env "$@" python3 bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > $O/$tag.json 2> $O/$tag.err; python3 -c "import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'])"; }
run base A=1
run geo_last PN2_GEO_FIRST=0
run nostreams PN2_MSG_STREAMS=0
run nostreams_geo_last PN2_MSG_STREAMS=0 PN2_GEO_FIRST=0
run mainlast0 PN2_MSG_MAIN_LAST=0
python3 bench.py --no-prefetch --no-cpu-baseline --no-roofline --steps 30 --warmup 5 > $O/noprefetch.json 2>$O/noprefetch.err; python3 -c "import json; d=json.loads(open('$O/noprefetch.json').read().strip().splitlines()[-1]); print('noprefetch', d['ms_per_step'])"
run base2 A=1
