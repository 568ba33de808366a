TWK_LIB=build/lib_rgdbg.so TWK_TRACE_REGROUP=1 timeout -k 10 120 python3 bench.py --steps 20 --warmup 0 --no-cpu-baseline --no-roofline 2>&1 | grep "^rg depth" | head -24
