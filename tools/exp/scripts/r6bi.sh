#!/bin/bash
# does a shape that never fails (H = 800, 13 k-blocks per wave) fail once the poll's M0 write sits right behind a state request?
export TMPDIR=/tmp
cd /root/repo
E=danspeech_amd/lib/libdsmi_exp.so
for sk in 0 256 512 768 1536; do
  echo "--- DSMI_DEBUG_RING_SKIP=$sk (poll's first read at k-block $((sk / 256)); 0: where the kernel has it, 7)"
  DSMI_LIBRARY=$E DSMI_RNN_KERNEL=ring4 DSMI_DEBUG_RING_SKIP=$sk timeout 600 python3 tools/exp/ring4_race.py 800 4 400 2>/dev/null | tail -2
done
