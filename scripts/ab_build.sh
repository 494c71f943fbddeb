#!/bin/bash
# development helper: builds ab_libs/libccm_hip_<name>.so from the working tree with ba.hip patched by a sed expression.  usage: ab_build.sh <name> '<sed expr>'
set -e
name=$1; expr=$2
rm -rf /tmp/ab_$name && mkdir -p /tmp/ab_$name/x /tmp/ab_$name/include
cp -r /root/repo/ccm_slam_amd/csrc /tmp/ab_$name/x/csrc && cp -r /root/repo/ccm_slam_amd/host /tmp/ab_$name/x/host && cp /root/repo/include/* /tmp/ab_$name/include/
cd /tmp/ab_$name/x/csrc && sed -i "$expr" ba.hip && touch ba.hip && make -j8 ../libccm_hip.so 2>&1 | grep -E "error|warning" || true
mkdir -p /root/repo/ab_libs && cp /tmp/ab_$name/x/libccm_hip.so /root/repo/ab_libs/libccm_hip_$name.so && ls -la /root/repo/ab_libs/libccm_hip_$name.so
