#!/bin/sh
# Regenerates the data fixtures of tests/golden from the reference tree (build container only;
# /root/reference does not exist on the GPU box and nothing at test time reads it).
set -e
cd "$(dirname "$0")"
for f in ml100k-train.csr ml100k-test.csr AutomotiveTrain.ijv AutomotiveTest.ijv l12file; do
  cp /root/reference/test/$f ./$f
  chmod 644 ./$f
done
