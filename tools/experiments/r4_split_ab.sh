bash tools/experiments/r4_split.sh 2>&1 | grep -v "^== quadtree_split 0" 
bash tools/experiments/r4_split_prof.sh 2>&1 | grep -E "^==|k_quadtree|k_qt_leaves"
bash tools/experiments/r4_qtp_timing.sh 2>&1 | grep -E "batch|k_qt_leaves"
