for r in 1 2; do for f in 1 0; do echo "FUSE_CQ=$f"; AX_WHISPER_FUSE_CQ=$f python profiles/scripts/ab_bench64.py 64; done; done
