set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03t
for m in ${MODES:-0 1 2}; do
  SFX_FUSE_TICK=$m timeout -k 10 300 python tests/tick_worker.py > gpurun_out/r03t/w$m.json 2> gpurun_out/r03t/w$m.err
  echo mode $m done
done
python - <<PY
import json
ms = "${MODES:-0 1 2}".split()
w = {m: json.load(open(f"gpurun_out/r03t/w{m}.json")) for m in ms}
for m in ms[1:]:
    print(m, "EQUAL" if w[m]["per_world"] == w[ms[0]]["per_world"] else "DIFFER", w[m]["tick_launches"])
PY
