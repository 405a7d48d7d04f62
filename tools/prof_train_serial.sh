#!/bin/bash
# as prof_train.sh with the weight gradients on the main stream (FSVIT_WGRAD_SIDE_STREAM=0): every kernel's duration is its duration alone on the GPU
# usage: bash tools/prof_train_serial.sh <tag> [extra bench args]
export FSVIT_WGRAD_SIDE_STREAM=0
exec bash "$(dirname "$0")/prof_train.sh" "$@"
