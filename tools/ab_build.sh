#!/bin/bash
# builds the committed HEAD as sota_imagenet_amd/lib/variant_old.so and the working tree as variant_new.so
set -e
cd "$(dirname "$0")/.."
make -C sota_imagenet_amd/csrc -j8 >/dev/null
cp sota_imagenet_amd/lib/libmi355rn.so /tmp/_ab_new.so
rm -rf /tmp/_ab_old && mkdir -p /tmp/_ab_old && git archive HEAD sota_imagenet_amd/csrc include | tar -x -C /tmp/_ab_old
make -C /tmp/_ab_old/sota_imagenet_amd/csrc -j8 >/dev/null
cp /tmp/_ab_old/sota_imagenet_amd/lib/libmi355rn.so sota_imagenet_amd/lib/variant_old.so
cp /tmp/_ab_new.so sota_imagenet_amd/lib/variant_new.so
ls -la sota_imagenet_amd/lib/
