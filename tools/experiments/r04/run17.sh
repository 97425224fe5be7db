#!/bin/bash
# serial schedule, postnet handed over in groups (each group's postnet on the side stream under the decoder's remaining steps); side stream priority
O=gpurun_out/r04_run17; mkdir -p $O
for cfg in "1 0" "4 0" "8 0" "4 1" "8 1" "16 1"; do set -- $cfg
echo "POSTNET_GROUPS=$1 SIDE_PRIO_LOW=$2"; VTTS_NAT_SL=1 VTTS_NAT_AHEAD=0 VTTS_PIPE_POSTNET_GROUPS=$1 VTTS_NAT_SIDE_PRIO=$2 timeout 300 python tools/pipeline_bench.py 256 1 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.endswith('_ms')})"; done
