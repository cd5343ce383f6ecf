# Makefile -- same targets and variables as the reference's (reference Makefile:3-26:
# EXE, all, check, clean, FINAL_STATE_FILE/AV_VELS_FILE/REF_*), building the
# MI355X-native program: a C host (gcc) on top of the HIP library (hipcc, gfx950).

EXE=d2q9-bgk

CC=gcc
HIPCC=hipcc
CFLAGS= -std=c99 -Wall -O2 -fopenmp
# -fno-slp-vectorize: the SLP vectoriser packs pairs of float operations into v_pk_* instructions, which on
# gfx950 are no faster than the two scalar ones and cost v_mov's to assemble their operands
# (8192^2, same box: lbm_march 237 -> 269 GLUPS, lbm_wave<8> 227 -> 262, lbm_sweep2 at 1024^2 132 -> 135)
HIPFLAGS= -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function -Wno-align-mismatch -fno-slp-vectorize
PKG=advanced-hpc-lbm_amd
LIB=$(PKG)/liblbm_mi355x.so

FINAL_STATE_FILE=./final_state.dat
AV_VELS_FILE=./av_vels.dat
REF_FINAL_STATE_FILE=tests/golden/1024x1024.final_state.pressure.f64.npz
REF_AV_VELS_FILE=tests/golden/1024x1024.av_vels.dat

all: $(EXE)

lib: $(LIB)

$(LIB): $(PKG)/csrc/lbm_api.hip $(PKG)/csrc/lbm_host_slabs.inc $(PKG)/csrc/lbm_host_march.inc $(PKG)/csrc/lbm_host_run.inc $(PKG)/csrc/lbm_kernels.hip.h $(PKG)/csrc/lbm_exact_math.hip.h $(PKG)/csrc/lbm_march.hip.h $(PKG)/csrc/lbm_wave.hip.h $(PKG)/csrc/lbm_regtile.hip.h include/lbm_mi355x.h
	$(HIPCC) $(HIPFLAGS) -shared $< -o $@ -ldl -Wl,-rpath,/opt/rocm/lib

$(EXE): $(PKG)/host/d2q9-bgk.c $(LIB) include/lbm_mi355x.h
	$(CC) $(CFLAGS) -Iinclude $< -o $@ -L$(PKG) -llbm_mi355x -Wl,-rpath,'$$ORIGIN/$(PKG)' -Wl,-rpath,/opt/rocm/lib

tools: tools/kbench tools/layout_bench tools/exact_math_check tools/oob_store_order tools/valu_chain tools/valu_forms

tools/valu_chain: tools/valu_chain.hip
	$(HIPCC) -O3 --offload-arch=gfx950 -fno-slp-vectorize $< -o $@

tools/valu_forms: tools/valu_forms.hip
	$(HIPCC) -O3 --offload-arch=gfx950 -fno-slp-vectorize $< -o $@

tools/oob_store_order: tools/oob_store_order.hip
	$(HIPCC) -O2 --offload-arch=gfx950 -Wno-unused-value $< -o $@

tools/layout_bench: tools/layout_bench.hip
	$(HIPCC) $(HIPFLAGS) $< -o $@

tools/exact_math_check: tools/exact_math_check.hip $(PKG)/csrc/lbm_exact_math.hip.h
	$(HIPCC) $(HIPFLAGS) $< -o $@

tools/kbench: tools/kbench.hip $(PKG)/csrc/lbm_kernels.hip.h $(PKG)/csrc/lbm_exact_math.hip.h
	$(HIPCC) $(HIPFLAGS) $< -o $@

oracle:
	$(MAKE) -C oracle all

check:
	python check/check_results.py --ref-av-vels-file=$(REF_AV_VELS_FILE) --ref-final-state-file=$(REF_FINAL_STATE_FILE) --av-vels-file=$(AV_VELS_FILE) --final-state-file=$(FINAL_STATE_FILE)

.PHONY: all lib tools oracle check clean

clean:
	rm -f $(EXE) $(LIB) tools/kbench tools/layout_bench tools/exact_math_check tools/oob_store_order tools/valu_chain tools/valu_forms
