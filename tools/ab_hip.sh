# A/B/A of two versions of csrc/attention.hip on ONE box: the tree's build, then tools/probe/att_prev.hip.txt (a copy of the other
# version, e.g. `git show HEAD~1:masters-thesis_amd/csrc/attention.hip`) built in its place, then the tree's again.  Separate
# processes drift by ~1 % even on one box (the two tree runs bracket that): differences below it need an in-process switch
# (tools/ab_attr.py).
set -e
R=$GRAFT_REPO_ROOT
cd $R
python tools/ab_attr.py attention | tail -1
cp masters-thesis_amd/csrc/attention.hip /tmp/att_new.hip
cp tools/probe/att_prev.hip.txt masters-thesis_amd/csrc/attention.hip
(cd masters-thesis_amd/csrc && make > /tmp/make.log 2>&1)
python tools/ab_attr.py attention | tail -1
cp /tmp/att_new.hip masters-thesis_amd/csrc/attention.hip
(cd masters-thesis_amd/csrc && make > /tmp/make.log 2>&1)
python tools/ab_attr.py attention | tail -1
