#!/bin/bash
# gpurun wrapper: stamps the snapshot with the commit it was taken at (.source_commit, git-ignored but shipped: the GPU box has
# no .git) - "<HEAD>" or "<HEAD>+dirty" - so that measurements copied into profiles/ can name their source state.
#   tools/gpurun.sh <timeout-seconds> '<command>'
cd /root/repo
c=$(git rev-parse --short=12 HEAD)
[ -n "$(git status --porcelain --untracked-files=no)" ] && c="$c+dirty"
echo "$c" > .source_commit
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
