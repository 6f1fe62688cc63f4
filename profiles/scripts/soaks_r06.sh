#!/bin/bash
# Soaks of the round-6 build (one gpurun call): every id sequence of one path against another path of the same build
O=gpurun_out/soaks6.txt
echo "# soaks of round 6 (MI355X)" > $O
echo "## profiles/scripts/soak_persistent.py 80 small" >> $O
timeout -k 10 300 python profiles/scripts/soak_persistent.py 80 small 2>&1 | tail -3 >> $O
echo "## profiles/scripts/soak_persistent2.py 40 small 2" >> $O
timeout -k 10 300 python profiles/scripts/soak_persistent2.py 40 small 2 2>&1 | tail -1 >> $O
for B in 64 5 11 19; do
  echo "## profiles/scripts/soak_batched.py $B small (one-branch steps: the clip-block query fold of round 6)" >> $O
  timeout -k 10 300 python profiles/scripts/soak_batched.py $B small 2>&1 | tail -5 >> $O
done
echo "## profiles/scripts/soak_stream.py small 384 16,64" >> $O
timeout -k 10 400 python profiles/scripts/soak_stream.py small 384 16,64 2>&1 | tail -3 >> $O
echo "## profiles/scripts/soak_server.py micro 600 24 16 (whisper_srv over HTTP)" >> $O
timeout -k 10 300 python profiles/scripts/soak_server.py micro 600 24 16 2>&1 | tail -3 >> $O
cat $O
