run() { ms=$(env "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])"); echo "$ms"; }
for i in 1 2 3; do
echo "sharded(default)      $(run python3 bench.py --emulate-world 8 --steps 192 --warmup 32 --no-cpu-baseline)"
echo "replicated one Adam   $(run python3 bench.py --emulate-world 8 --steps 192 --warmup 32 --no-cpu-baseline --replicated-update)"
echo "replicated group pipe $(run env LRCN_DP_GROUP_ADAM=1 python3 bench.py --emulate-world 8 --steps 192 --warmup 32 --no-cpu-baseline --replicated-update)"
echo "repl group+sparse     $(run env LRCN_DP_GROUP_ADAM=1 LRCN_DP_SPARSE_EMBED=1 python3 bench.py --emulate-world 8 --steps 192 --warmup 32 --no-cpu-baseline --replicated-update)"
done
