#!/bin/bash
# Dev tool (container): build HEAD's library into tools/lab/libvneti_prev.so (the A/B baseline of step_ab.sh / halo_ab.sh),
# then rebuild the working tree's.
cd /root/repo || exit 1
git stash -q || exit 1
python view_neti_amd/csrc/build.py 2>&1 | grep "libvneti_hip.so"
cp view_neti_amd/csrc/libvneti_hip.so tools/lab/libvneti_prev.so
git stash pop -q
python view_neti_amd/csrc/build.py 2>&1 | grep built
