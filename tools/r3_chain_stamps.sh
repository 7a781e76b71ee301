# builds tools/chain_stamps from stamped copies of the engine's kernels (no library needed) — run from the repo root
set -e
C=qwen3-tts-rust_amd/csrc
F="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -w -DQ3_STAMPS -I $C"
mkdir -p tools/exp/bs
hipcc $F -c $C/q3_bgemm.hip -o tools/exp/bs/cs_bgemm.o &
hipcc $F -c $C/q3_kernels.hip -o tools/exp/bs/cs_kernels.o &
hipcc $F -c tools/chain_stamps.hip -o tools/exp/bs/cs_main.o &
wait
hipcc --offload-arch=gfx950 -o tools/chain_stamps tools/exp/bs/cs_main.o tools/exp/bs/cs_bgemm.o tools/exp/bs/cs_kernels.o
