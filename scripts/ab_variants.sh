# A/B of experiment builds (cam-nor-physics_amd/_build/var/lib_<name>.so): bash scripts/ab_variants.sh name1 name2 ... [-- bench args]
names=(); extra=()
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; extra=("$@"); break; fi; names+=("$1"); shift; done
for v in "${names[@]}"; do
  COLPHYS_LIB=cam-nor-physics_amd/_build/var/lib_$v.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline "${extra[@]}" > gpurun_out/var_$v.log 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/var_$v.log')); k=d['roofline']['kernel_ms']; print('$v', d['ms_per_step'], {a: round(b,3) for a,b in k.items()})"
done
