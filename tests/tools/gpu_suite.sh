#!/bin/bash
# Developer helper for gpurun: the whole -m gpu suite (one process), log under gpurun_out/<tag>/.
TAG=${1:-suite}
mkdir -p gpurun_out/$TAG
python -m pytest tests -m gpu -q ${@:2} > gpurun_out/$TAG/tests.log 2>&1
rc=$?
tail -6 gpurun_out/$TAG/tests.log
exit $rc
