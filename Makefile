# libgmr1_hip.so without Python: the MI355X-native GMR-1 receive path behind the reference's C API.
#
#   make                 osmo-gmr_amd/libgmr1_hip.so (hipcc, gfx950; objects under build/)
#   make check           tests/c/abi_smoke.c over every public header, linked against the library and run (no GPU needed)
#   make install         PREFIX/lib/libgmr1_hip.so, PREFIX/include/{gmr1_hip.h,gmr1_hip_shard.h,osmocom/gmr1/**},
#                        PREFIX/lib/pkgconfig/gmr1_hip.pc      (PREFIX=/usr/local, DESTDIR honoured)
#   make gmr1_rx REF=/path/to/osmo-gmr
#                        the reference's unchanged application (src/gmr1_rx.c + src/gsmtap.c, reference src/Makefile.am:1-24)
#                        against this library and the system's libosmocore / libosmodsp (pkg-config); what the reference
#                        links as libgmr1-sdr / libgmr1-l1 (src/sdr/Makefile.am:5-7, src/l1/Makefile.am:5-9) is this one .so
#   make profile         osmo-gmr_amd/libgmr1_hip_prof.so (-DGMR1_HIP_PROFILE: the debug / measurement switches)
#
# The same sources, flags and output as osmo-gmr_amd/build.py (which the Python tests and bench.py use); either may be
# run after the other.  No CPU fallback is built: without a HIP device every compute entry point returns -ENODEV.

HIPCC    ?= $(shell command -v hipcc 2>/dev/null || echo /opt/rocm/bin/hipcc)
ARCH     ?= gfx950
PREFIX   ?= /usr/local
LIBDIR   ?= $(PREFIX)/lib
INCDIR   ?= $(PREFIX)/include
VERSION  := 0.4
BUILD    ?= build

CSRC     := osmo-gmr_amd/csrc
LIB      := osmo-gmr_amd/libgmr1_hip.so
PROFLIB  := osmo-gmr_amd/libgmr1_hip_prof.so

HIP_SRC  := rx_kernels.hip fcch_kernels.hip l1_kernels.hip tch_kernels.hip chan_kernels.hip nt9_kernels.hip \
            xch_kernels.hip tx_kernels.hip ambe_kernels.hip util_kernels.hip
CXX_SRC  := capi.cpp capi_fcch.cpp capi_l1.cpp capi_detect.cpp capi_rx.cpp capi_tch.cpp capi_chan.cpp capi_nt9.cpp \
            capi_xch.cpp capi_tx.cpp host_tables.cpp l1_tables.cpp l1_punct.cpp capi_shard.cpp capi_ambe.cpp ambe_tables.cpp

CXXFLAGS := -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Iinclude -I$(CSRC)
HEADERS  := $(wildcard $(CSRC)/*.h $(CSRC)/*.inc) $(shell find include -name '*.h')

OBJS     := $(addprefix $(BUILD)/,$(HIP_SRC:.hip=.o) $(CXX_SRC:.cpp=.o))
PROFOBJS := $(addprefix $(BUILD)/prof/,$(HIP_SRC:.hip=.o) $(CXX_SRC:.cpp=.o))

all: $(LIB)

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

profile: $(PROFLIB)

$(PROFLIB): $(PROFOBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(PROFOBJS)

$(BUILD)/%.o: $(CSRC)/%.hip $(HEADERS) | $(BUILD)
	$(HIPCC) -xhip $(CXXFLAGS) --offload-arch=$(ARCH) -c $< -o $@

$(BUILD)/%.o: $(CSRC)/%.cpp $(HEADERS) | $(BUILD)
	$(HIPCC) $(CXXFLAGS) --offload-arch=$(ARCH) -c $< -o $@

$(BUILD)/prof/%.o: $(CSRC)/%.hip $(HEADERS) | $(BUILD)/prof
	$(HIPCC) -xhip $(CXXFLAGS) -DGMR1_HIP_PROFILE --offload-arch=$(ARCH) -c $< -o $@

$(BUILD)/prof/%.o: $(CSRC)/%.cpp $(HEADERS) | $(BUILD)/prof
	$(HIPCC) $(CXXFLAGS) -DGMR1_HIP_PROFILE --offload-arch=$(ARCH) -c $< -o $@

$(BUILD) $(BUILD)/prof:
	mkdir -p $@

$(BUILD)/gmr1_hip.pc: Makefile | $(BUILD)
	@printf 'prefix=%s\nlibdir=%s\nincludedir=%s\n\nName: gmr1_hip\nDescription: MI355X-native GMR-1 receive path behind the libosmo-gmr1 C API (HIP, gfx950)\nVersion: %s\nLibs: -L$${libdir} -lgmr1_hip\nCflags: -I$${includedir}\n' \
		'$(PREFIX)' '$(LIBDIR)' '$(INCDIR)' '$(VERSION)' > $@

# every public header is plain C99, the structs have the documented sizes, the host-only calls run (no GPU needed)
check: $(LIB) | $(BUILD)
	$(CC) -std=c99 -Wall -Werror -Iinclude tests/c/abi_smoke.c -o $(BUILD)/abi_smoke \
		-Wl,--no-undefined -Losmo-gmr_amd -l:libgmr1_hip.so -Wl,-rpath,$(abspath osmo-gmr_amd) -Wl,-rpath,/opt/rocm/lib
	$(BUILD)/abi_smoke
	@echo "check: ok"

install: $(LIB) $(BUILD)/gmr1_hip.pc
	install -d $(DESTDIR)$(LIBDIR)/pkgconfig $(DESTDIR)$(INCDIR)
	install -m 755 $(LIB) $(DESTDIR)$(LIBDIR)/libgmr1_hip.so
	install -m 644 $(BUILD)/gmr1_hip.pc $(DESTDIR)$(LIBDIR)/pkgconfig/gmr1_hip.pc
	cd include && find . -name '*.h' | while read h; do install -D -m 644 "$$h" "$(DESTDIR)$(INCDIR)/$$h"; done

# the reference's application, unchanged (needs libosmocore / libosmodsp installed; REF = a checkout of osmo-gmr)
gmr1_rx: $(LIB)
	@test -n "$(REF)" || { echo "usage: make gmr1_rx REF=/path/to/osmo-gmr"; exit 2; }
	$(CC) -std=gnu99 -O2 -Wall -DGMR1_HIP_USE_SYSTEM_OSMOCOM -Iinclude $$(pkg-config --cflags libosmocore libosmodsp) \
		-o gmr1_rx $(REF)/src/gmr1_rx.c $(REF)/src/gsmtap.c \
		-Losmo-gmr_amd -l:libgmr1_hip.so -Wl,-rpath,$(abspath osmo-gmr_amd) $$(pkg-config --libs libosmocore libosmodsp) -lm

clean:
	rm -rf $(BUILD) gmr1_rx

.PHONY: all profile check install gmr1_rx clean
