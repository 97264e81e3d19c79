#!/bin/bash
# GPU box: the -m gpu suite in ONE process, then smoke(); logs under gpurun_out/
cd "$(dirname "$0")/../.."
TAG=${1:-suite}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; rc=$?
tail -15 gpurun_out/${TAG}_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1; rc=$?
tail -5 gpurun_out/${TAG}_smoke.log
exit $rc
