# A/B of builds of the library on one box: PK2_AB_LIBS="libpk2hip.so libpk2hip_x.so"; PK2_AB_REPS; PK2_AB_CFGS (; separated)
IFS=';' read -ra CFGS <<< "${PK2_AB_CFGS:-;--den-states 30000 --den-arcs 1500000}"
for rep in $(seq 1 ${PK2_AB_REPS:-2}); do
for lib in ${PK2_AB_LIBS:-libpk2hip.so}; do
  export PK2_LIB=$PWD/pykaldi2_amd/$lib
  line="$lib"
  for cfg in "${CFGS[@]}"; do
    v=$(python bench.py --den-only $cfg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['us_per_frame'])")
    line="$line | ${cfg:-bench graph} $v"
  done
  echo "$line"
done; done
