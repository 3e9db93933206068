#!/bin/bash
# The reference's literal multi-frame sequence (clear, 9 nonlocal.comp dispatches at [-7,7)/[-3,3), normalize) from compiled host code:
# and the PNG path of a single-image mode (unpack, bilateral r = 4, pack) for 9 frames = 28 short launches,
# issued call by call against one mid_recording_submit of the recorded calls (tests/recording_host.cpp).  -> profiles/r06_recording_replay.txt
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
L=image_denoising_filter_amd
g++ -std=c++17 -O2 -Wall -I include tests/recording_host.cpp -o /tmp/recording_host -L $L -lmi_denoise -Wl,-rpath,$PWD/$L -Wl,-rpath,/opt/rocm/lib
for pass in 1 2; do
  for size in "64 64 2000" "128 128 2000" "256 256 1000" "512 512 300" "1920 1080 20"; do
    set -- $size
    /tmp/recording_host $1 $2 9 $3 /tmp/recording_out.raw
    /tmp/recording_host $1 $2 9 $3 /tmp/recording_out.raw ldr
  done
done
