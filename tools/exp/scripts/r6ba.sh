#!/bin/bash
# beam searches of a batch as launches of n utterances (DSMI_TEST_BEAM_CHUNK): configs 3, 4, 5
export TMPDIR=/tmp
cd /root/repo
for c in 0 32 16; do
  echo "--- DSMI_TEST_BEAM_CHUNK=$c"
  DSMI_TEST_BEAM_CHUNK=$c python3 tools/exp/config_stream.py 3 4 64 2>/dev/null | tail -1
  DSMI_TEST_BEAM_CHUNK=$c python3 tools/exp/config_stream.py 3 4 64 2>/dev/null | tail -1
  DSMI_TEST_BEAM_CHUNK=$c python3 tools/exp/config_stream.py 4 2 32 2>/dev/null | tail -1
  DSMI_TEST_BEAM_CHUNK=$c python3 tools/exp/config_stream.py 5 4 12 2>/dev/null | tail -1
done
DSMI_TEST_BEAM_CHUNK=16 timeout 900 python3 -m pytest tests/test_gpu_beam.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
