# Experiment builds with in-kernel 100 MHz timestamps (q3_kernels.h Q3_STAMP); run from the repo root, here or on the GPU box.
#   bash tools/build_stamps.sh chain   -> tools/chain_stamps (a chain of the engine's own kernels, built with -DQ3_STAMPS; no library needed)
#   bash tools/build_stamps.sh voc     -> tools/exp/libq3tts_vstamps.so (libq3tts with -DQ3_VOC_STAMPS in q3_vocoder.hip only; needs `make` first)
set -e
C=qwen3-tts-rust_amd/csrc
mkdir -p tools/exp/bs
case ${1:-chain} in
chain)
  F="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -w -DQ3_STAMPS -I $C"
  hipcc $F -c $C/q3_bgemm.hip -o tools/exp/bs/cs_bgemm.o &
  hipcc $F -c $C/q3_kernels.hip -o tools/exp/bs/cs_kernels.o &
  hipcc $F -c tools/chain_stamps.hip -o tools/exp/bs/cs_main.o &
  wait
  hipcc --offload-arch=gfx950 -o tools/chain_stamps tools/exp/bs/cs_main.o tools/exp/bs/cs_bgemm.o tools/exp/bs/cs_kernels.o;;
voc)
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -w -DQ3_VOC_STAMPS -c $C/q3_vocoder.hip -o tools/exp/bs/vs_vocoder.o
  O=$(ls $C/build/*.o | grep -v q3_vocoder)
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libq3tts_vstamps.so $O tools/exp/bs/vs_vocoder.o -ldl -lpthread;;
esac
