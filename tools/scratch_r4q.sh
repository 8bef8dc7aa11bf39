for v in "" a2f10 a2b10 a2f12 ""; do echo "variant '$v':"; SAVIT_EXP_LIB=$v python tools/attn_bench.py 256 577 16 64 2>&1 | grep -v amdgpu; done
