# Builds the kernel library with the in-kernel phase timestamps of the attention model's chains (-DTNT_LC_TRACE) OUT OF TREE,
# into .trace_build/ (git-ignored, travels with gpurun); run tools/lc_trace.py with TNT_HIP_LIB pointing at it:
#   sh tools/lc_trace_build.sh && gpurun -- 'TNT_HIP_LIB=$PWD/.trace_build/csrc/libtnt_hip.so python tools/lc_trace.py'
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/.trace_build/include
rm -rf $R/.trace_build/pkg; mkdir -p $R/.trace_build/pkg
cp -r $R/masters-thesis_amd/csrc $R/.trace_build/pkg/csrc
cp $R/include/tnt_hip.h $R/.trace_build/include/
cd $R/.trace_build/pkg/csrc && rm -f attention.o lstm.o libtnt_hip.so
make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DTNT_LC_TRACE" attention.o lstm.o > ../make.log 2>&1
make >> ../make.log 2>&1
ls -la libtnt_hip.so
