# usage (GPU box): bash tools/_abenv.sh VAR "v1 v2" "cfg1 cfg2" [reps] — interleaved same-box A/B of one environment switch
var=$1; vals=$2; cfgs=${3:-"c5hoi c5hhi"}; reps=${4:-3}
for rep in $(seq $reps); do for c in $cfgs; do for v in $vals; do
  env $var=$v python3 bench.py --config $c --no-cpu-baseline --no-roofline --min-seconds 1.2 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$c $var=$v', round(j['ms_per_step'],4))"
done; done; done
