# Build of the MI355X `MethylDackel extract` path: device library (hipcc, gfx950), host library (gcc),
# the `MethylDackel` command, plus the test/bench infrastructure (oracle, synthetic generator).
HIPCC  ?= hipcc
CC     ?= gcc
ARCH   ?= gfx950
HIPFLAGS ?=
B      := methyldackel_amd/_build
CFLAGS ?= -O2 -g -Wall -Wextra -Wno-unused-parameter -Wno-sign-compare -fPIC -pthread
HOSTSRC := methyldackel_amd/csrc/host/mdk_io.c methyldackel_amd/csrc/host/mdk_fasta.c methyldackel_amd/csrc/host/mdk_bigwig.c methyldackel_amd/csrc/host/mdk_mbias.c methyldackel_amd/csrc/host/mdk_mergecontext.c methyldackel_amd/csrc/host/mdk_plan.c methyldackel_amd/csrc/host/mdk_pipeline.c methyldackel_amd/csrc/host/mdk_emit.c methyldackel_amd/csrc/host/mdk_extract.c methyldackel_amd/csrc/host/mdk_cmd_mbias.c methyldackel_amd/csrc/host/mdk_cmd_perread.c methyldackel_amd/csrc/host/mdk_affinity.c methyldackel_amd/csrc/host/mdk_ranks.c

all: $(B)/libmdk_hip.so $(B)/libmdk_extract.so $(B)/MethylDackel tools oracle

# -fgpu-rdc: the four sources become ONE code object; the runtime loads a code object at the first use of one of its kernels and each load
# costs the command ~30 ms of start-up (three of them did: pileup, preparation, inflate)
HIPSRC := methyldackel_amd/csrc/mdk_hip.hip methyldackel_amd/csrc/mdk_comm.hip methyldackel_amd/csrc/mdk_prep.hip methyldackel_amd/csrc/mdk_inflate.hip
$(B)/libmdk_hip.so: $(HIPSRC) methyldackel_amd/csrc/mdk_hip_internal.hpp methyldackel_amd/csrc/mdk_overlap_rule.h methyldackel_amd/csrc/mdk_pair_rule.h methyldackel_amd/csrc/mdk_inflate_core.h methyldackel_amd/csrc/mdk_crc32_core.h include/mdk_hip.h
	@mkdir -p $(B)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -fgpu-rdc $(HIPFLAGS) -Iinclude -Imethyldackel_amd/csrc -o $@ $(HIPSRC) -ldl

$(B)/libmdk_extract.so: $(HOSTSRC) methyldackel_amd/csrc/host/mdk_io.h methyldackel_amd/csrc/host/mdk_plan.h include/mdk_extract.h include/mdk_hip.h $(B)/libmdk_hip.so
	$(CC) $(CFLAGS) -shared -Iinclude -o $@ $(HOSTSRC) -L$(B) -lmdk_hip -Wl,-rpath,'$$ORIGIN' -lz -lm

$(B)/MethylDackel: methyldackel_amd/csrc/host/main.c $(B)/libmdk_extract.so
	$(CC) $(CFLAGS) -Iinclude -o $@ methyldackel_amd/csrc/host/main.c -L$(B) -lmdk_extract -lmdk_hip -Wl,-rpath,'$$ORIGIN' -lz -lm

tools: tools/_build/mdk_synth tools/_build/mdk_replicate tools/_build/fasta_probe tools/_build/mdk_calib tools/_build/inflate_emu tools/_build/piece_bench tools/_build/pin_probe tools/_build/feed_harness tools/_build/libmdk_piece_standin.so tools/_build/libmdk_dev_standin.so
tools/_build/libmdk_piece_standin.so: tools/piece_standin.c include/mdk_hip.h
	@mkdir -p tools/_build
	$(CC) -O2 -g -Wall -shared -fPIC -Iinclude -o $@ tools/piece_standin.c -lz
tools/_build/libmdk_dev_standin.so: tools/dev_standin.c tools/piece_standin.c include/mdk_hip.h
	@mkdir -p tools/_build
	$(CC) -O2 -g -Wall -shared -fPIC -Iinclude -Itools -o $@ tools/dev_standin.c -lz -lpthread
tools/_build/feed_harness: tools/feed_harness.c methyldackel_amd/csrc/host/mdk_io.c methyldackel_amd/csrc/host/mdk_io.h include/mdk_hip.h
	@mkdir -p tools/_build
	$(CC) -O2 -g -Wall -Iinclude -Imethyldackel_amd/csrc/host -o $@ tools/feed_harness.c -lz -lpthread -ldl
tools/_build/pin_probe: tools/pin_probe.hip
	@mkdir -p tools/_build
	$(HIPCC) --offload-arch=$(ARCH) -O2 -o $@ tools/pin_probe.hip
tools/_build/inflate_emu: tools/inflate_emu.cpp methyldackel_amd/csrc/mdk_inflate_core.h methyldackel_amd/csrc/mdk_crc32_core.h
	@mkdir -p tools/_build
	g++ -O2 -Wall -Wno-unknown-pragmas -o $@ tools/inflate_emu.cpp -Imethyldackel_amd/csrc -lz
tools/_build/piece_bench: tools/piece_bench.c include/mdk_hip.h $(B)/libmdk_hip.so
	@mkdir -p tools/_build
	$(CC) -O2 -g -Wall -Iinclude -o $@ tools/piece_bench.c -L$(B) -lmdk_hip -Wl,-rpath,'$$ORIGIN/../../$(B)' -lz
tools/_build/mdk_calib: tools/mdk_calib.hip
	@mkdir -p tools/_build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ tools/mdk_calib.hip
tools/_build/fasta_probe: tools/fasta_probe.c methyldackel_amd/csrc/host/mdk_fasta.c methyldackel_amd/csrc/host/mdk_io.h
	@mkdir -p tools/_build
	$(CC) -O2 -g -Wall -Iinclude -o $@ tools/fasta_probe.c methyldackel_amd/csrc/host/mdk_fasta.c -lpthread
tools/_build/mdk_replicate: tools/mdk_replicate.c
	@mkdir -p tools/_build
	$(CC) -O2 -g -Wall -o $@ tools/mdk_replicate.c -lz -lpthread
tools/_build/mdk_synth: tools/mdk_synth.c
	@mkdir -p tools/_build
	$(CC) -O2 -g -o $@ tools/mdk_synth.c -lz -lm -lpthread

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(B) tools/_build oracle/_build
.PHONY: all tools oracle clean
