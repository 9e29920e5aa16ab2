#!/bin/bash
# repeated A/B (timing noise between processes is large): tools/ab_rep.sh "<lib specs>" <repeats>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
libs=$1; n=${2:-3}; shift; shift
for r in $(seq 1 $n); do bash tools/ab.sh "$libs" "$@" 2>&1 | grep -o "^==.*\|^[0-9.]* \|'scatter1': [0-9.]*\|'scatter2': [0-9.]*\|'count': [0-9.]*\|'sort': [0-9.]*" | paste -sd' ' | sed 's/==/\n==/g'; done
