set -e
mkdir -p gpurun_out/r3
for occ in 0 6; do
  MP_SEED_OCC=$occ python3 bench.py --steps 5 --warmup 2 --no-pmc --no-cpu-baseline --no-legs --contexts 1 > gpurun_out/r3/occ_$occ.json 2> gpurun_out/r3/occ_$occ.err || { tail -20 gpurun_out/r3/occ_$occ.err; exit 1; }
  python3 -c "
import json; j=json.load(open('gpurun_out/r3/occ_$occ.json')); print('occ $occ 1ctx', j['value'], j['kernel_ms_per_step']['seed'])"
  MP_SEED_OCC=$occ python3 bench.py --steps 8 --warmup 2 --no-pmc --no-cpu-baseline --no-legs > gpurun_out/r3/occ3_$occ.json 2> gpurun_out/r3/occ3_$occ.err
  python3 -c "
import json; j=json.load(open('gpurun_out/r3/occ3_$occ.json')); print('occ $occ 3ctx', j['value'], j['ms_per_step'])"
done
MP_KMER_FILTER=0 MP_KMER15=0 MP_LKT16=0 MP_SA_DENSE=0 MP_SEED_OCC=6 python3 bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-legs --contexts 1 --ini soap4-nt2.ini > gpurun_out/r3/occ_lean6.json 2>/dev/null
MP_KMER_FILTER=0 MP_KMER15=0 MP_LKT16=0 MP_SA_DENSE=0 python3 bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-legs --contexts 1 --ini soap4-nt2.ini > gpurun_out/r3/occ_lean0.json 2>/dev/null
python3 -c "
import json
for t in ('lean0','lean6'):
    j=json.load(open('gpurun_out/r3/occ_%s.json'%t)); print(t, j['value'], j['kernel_ms_per_step']['seed'], j['kernel_ms_per_step']['locate'])"
