#!/bin/bash
# round 6, call u: the dense token in the product library: config 2 on / off, the other configs on / off, the whole GPU suite
export TMPDIR=/tmp
cd /root/repo
for T in 1 0; do
  echo "=== DSMI_DENSE_TOKENS=$T"
  DSMI_DENSE_TOKENS=$T python3 tools/exp/short_calls.py 4 8 20 96 2>&1 | grep "batches per call" | cut -c1-100
  DSMI_DENSE_TOKENS=$T python3 tools/exp/config_stream.py 3 4 64 2>&1 | grep "^config"
  DSMI_DENSE_TOKENS=$T python3 tools/exp/config_stream.py 4 2 48 2>&1 | grep "^config"
  DSMI_DENSE_TOKENS=$T python3 tools/exp/config_stream.py 5 4 16 2>&1 | grep "^config"
done
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
