#!/bin/bash
# per-round host-side timing of the commit chain (STARK_HIP_DEBUG=1) at 2^20 and for the 128 x 2^14 batch
cd "$(dirname "$0")/.."
STARK_HIP_DEBUG=1 timeout -k 5 60 python tools/bench_commit.py stark_anatomy_amd/libstark_hip.so 20 2>&1 | tail -40
STARK_HIP_DEBUG=1 timeout -k 5 60 python tools/bench_batch.py 128 14 2>&1 | tail -30
