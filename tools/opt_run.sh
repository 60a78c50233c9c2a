# bench.py under a list of engine options (OPTS="a=1 b=2;c=3": one run per ';'-separated set), one GPU session
mkdir -p gpurun_out/opt
IFS=';' read -ra SETS <<< "${OPTS}"
for rep in 1 2; do
for set in "${SETS[@]}"; do
  args=""; for o in $set; do args="$args --opt $o"; done
  tag=$(echo "$set" | tr ' =' '__')
  timeout -k 10 200 python3 bench.py --box ${BOX:-64} --steps ${STEPS:-1500} --warmup 200 --no-cpu-baseline --profile-steps 50 $args > gpurun_out/opt/$tag.json 2>gpurun_out/opt/err.txt || exit 1
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/opt/$tag.json").read().strip().splitlines()[-1])
r=d["roofline"]; p=d["phases_ms"]
print("%-28s %.0f steps/s  pair %.1f us (only %.1f)  neigh %.0f reorder %.0f bin %.0f" % ("$set", d["value"], (r.get("fused") or r)["us_per_launch"], r["us_per_launch"], p["neigh"]*1e3, p["reorder"]*1e3, p["bin"]*1e3))
PY
done
done
