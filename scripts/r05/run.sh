#!/bin/bash
# local wrapper: rebuild the library (and the attention ubench variants) so that the box never sees a stale .so, then one gpurun call
# usage: scripts/r05/run.sh <timeout-seconds> <script on the box>
cd /root/repo
python -c "from gvfdiffusion_amd import _build; _build.build()" 2>&1 | grep -v packed-fp32
/usr/local/graft/bin/gpurun --timeout ${1:-1800} -- "bash $2"
