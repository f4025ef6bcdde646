#!/bin/bash
# tools/gpu_round4_final.sh: the round's final evidence on the final code in one call: what the driver does (tests, smoke, bench at its flags), then the profile set
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
tools/gpu_job.sh check r04z
tools/collect_round.sh r04z
