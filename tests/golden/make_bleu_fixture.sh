#!/bin/sh
# Generates tests/golden/bleu/*: the first 200 lines of the reference's committed COCO candidates + 5 reference files
# (data files of /root/reference/eval, not source) and the expected output of the reference's own multi-bleu.perl on them,
# plus the script's output on the four full known-answer sets of SURVEY.md section 4.  Run in the build container only.
set -e
R=/root/reference/eval
D=$(dirname "$0")/bleu
mkdir -p "$D"
head -200 $R/candidates.txt > $D/cand200.txt
for i in 0 1 2 3 4; do head -200 $R/coco_refs/ref$i > $D/ref200_$i; done
( cd $D && for i in 0 1 2 3 4; do cp ref200_$i r$i; done; perl $R/multi-bleu.perl ./r < cand200.txt > expected200.txt; rm -f r0 r1 r2 r3 r4 )
( cd $R && perl multi-bleu.perl ./coco_refs/ref < candidates.txt; perl multi-bleu.perl ./flickr_refs/f_ref < caps_flickr_bm3; \
  perl multi-bleu.perl ./flickr_refs/f_ref < caps_flickr_bm5; perl multi-bleu.perl ./flickr_refs/f_ref < caps_flickr_bm10 ) > $D/expected_full.txt
