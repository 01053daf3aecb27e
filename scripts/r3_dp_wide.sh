# full-table kernel with traceback: its launches are small on the default workload (the band tiers' left-overs, the passing rescue instances):
# 16 lanes x 10 columns per instance pair (default) against 32 x 5 (MP_DP_WIDE=1: four times shorter critical path per wavefront)
set -e
mkdir -p gpurun_out/r3
for v in "base" "MP_DP_WIDE=1"; do
  if [ "$v" = base ]; then python3 bench.py --no-legs --no-pmc --no-cpu-baseline --steps 5 > gpurun_out/r3/dpwide_$v.json 2> gpurun_out/r3/dpwide.err
  else env $v python3 bench.py --no-legs --no-pmc --no-cpu-baseline --steps 5 > gpurun_out/r3/dpwide_$v.json 2> gpurun_out/r3/dpwide.err; fi
  python3 -c "
import json,sys; j=json.load(open('gpurun_out/r3/dpwide_$v.json'))
k=j['kernel_ms_per_step']
print('$v', round(j['value']/1e6,2), 'M pairs/s; one context', round(j['single_context']['ms_per_step'],2), 'fill', round(k['fill'],2), 'trace', round(k['trace'],2), 'rescue_fill', round(k['rescue_fill'],2), 'rescue_trace', round(k['rescue_trace'],2))"
done
