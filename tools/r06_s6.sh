#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06_s6
timeout 120 tools/micro/count_tag > gpurun_out/r06_s6/count_tag.txt 2>&1; tail -10 gpurun_out/r06_s6/count_tag.txt
