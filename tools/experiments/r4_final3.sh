#!/bin/bash
# a second box for the final build's bench lines (box-to-box spread of K2 is ~5 %)
export TMPDIR=/tmp
timeout 1500 bash tools/baseline_table.sh r04_end2
SUSTAIN=45 EVERY=6 timeout 400 bash tools/sustained_run.sh r04_end2 | head -4
