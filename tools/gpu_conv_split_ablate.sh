for S in 1 257 513 1025 2049 3073 3841 0; do echo -n "split=$S "; AIRGYM_EXPERIMENTS=1 AIRGYM_CONV_SPLIT=$S python tools/conv_probe.py --layers conv2 2>/dev/null | head -1 | cut -c1-100; done
