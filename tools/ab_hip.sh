# A/B/A of two versions of csrc/attention.hip on ONE box: the tree's build, then tools/probe/att_prev.hip.txt (a copy of the other
# version, e.g. `git show HEAD~1:masters-thesis_amd/csrc/attention.hip`) built OUT OF TREE in a scratch copy of csrc/ and loaded
# through the TNT_HIP_LIB override of _lib.py, then the tree's again.  The work tree is never modified, so a failed build or run
# cannot leave the wrong kernel source (or library) behind.  Separate processes drift by ~1 % even on one box (the two tree runs
# bracket that): differences below it need an in-process switch (tools/ab_attr.py).
set -e
R=$GRAFT_REPO_ROOT
cd $R
S=$(mktemp -d /tmp/ab_hip.XXXXXX)
trap 'rm -rf "$S"' EXIT
mkdir -p $S/pkg $S/include
cp -r masters-thesis_amd/csrc $S/pkg/csrc
cp include/tnt_hip.h $S/include/
cp tools/probe/att_prev.hip.txt $S/pkg/csrc/attention.hip
(cd $S/pkg/csrc && rm -f attention.o libtnt_hip.so && make > $S/make.log 2>&1)
python tools/ab_attr.py attention | tail -1
TNT_HIP_LIB=$S/pkg/csrc/libtnt_hip.so python tools/ab_attr.py attention | tail -1
python tools/ab_attr.py attention | tail -1
