#!/bin/bash
# conv_probe over the vocoder's layer shapes (8 utterances of 500 frames)
P=tools/_build/conv_probe
for spec in "8 256 256 3 1 4000 1" "8 256 256 7 1 4000 1" "8 256 256 11 5 4000 1" "8 128 128 3 1 20000 1" "8 128 128 11 1 20000 1" \
            "8 64 64 3 1 60001 1" "8 64 64 7 3 60001 1" "8 64 64 11 5 60001 1" "8 64 64 11 5 60001 0" "8 64 64 3 1 60001 0" \
            "32 64 64 3 1 60001 0" "8 1024 64 11 1 60001 0" "8 256 256 11 1 16000 0"; do
  timeout -k 5 60 $P $spec 20 || exit 1
done
