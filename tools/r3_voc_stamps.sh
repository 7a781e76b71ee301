# experiment build: libq3tts with -DQ3_VOC_STAMPS in q3_vocoder.hip only (tools/exp/libq3tts_vstamps.so) — run from the repo root after `make`
set -e
C=qwen3-tts-rust_amd/csrc
mkdir -p tools/exp/bs
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -w -DQ3_VOC_STAMPS -c $C/q3_vocoder.hip -o tools/exp/bs/vs_vocoder.o
O=$(ls $C/build/*.o | grep -v q3_vocoder)
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libq3tts_vstamps.so $O tools/exp/bs/vs_vocoder.o -ldl -lpthread
