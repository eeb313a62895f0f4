# A/B of libmnv.so variants built by tools/build_variant.sh: bash tools/ab_lib.sh <tag>[:ENV=VALUE] ...   ("base" = the regular build)
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  if [ "$v" = base ]; then unset MNV_LIB_PATH; else export MNV_LIB_PATH=$PWD/variants/libmnv_$v.so; fi
  env $envs python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$spec', d['value'], 'Mrays/s', d['ms_per_step'], 'ms; per_frame', d['per_frame']['value'])"
done
