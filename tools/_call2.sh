#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
LVX_LAYOUT_TIMING=1 python tools/probes/stage_times.py 2>&1 | grep "layout\|trajInit"
