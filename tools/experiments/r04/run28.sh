#!/bin/bash
# hoisted gates added after the K shares (prefetched per block) against accumulators that start from them: interleaved on one box
for rep in 1 2 3; do for v in libvtts_head.so libvtts_hifigan.so; do for cfg in fp32 x3; do
echo -n "$v $cfg  "; VTTS_HIFIGAN_LIB=$PWD/viettts_amd/lib/$v timeout 300 python tools/pipeline_bench.py 256 1 3 $cfg 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('acoustic_model_ms','total_ms')})"; done; done; done
