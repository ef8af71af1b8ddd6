for co in 256 320; do for h in 7 10 9 1; do HW=64 CI=320 CO=$co HINT=$h REPS=10 python tools/pmc_conv.py 2>/dev/null | tail -1; done; done
