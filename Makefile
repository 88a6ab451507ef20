# Plain-make build of the MI355X window-scan engine, for users who come from the reference's own Makefile
# (g++ -O3 on one file per tool, Makefile:1-27 there).  python -m popgenomicstools_amd.build does the same
# with staleness checks (a content hash beside the library: after a `make` of changed sources the Python
# loader rebuilds once more and re-stamps); both leave libpgtwin.so and the five host tools in the same places.
#
#   make            libpgtwin.so + bin/{fstWindow,hetWindow,dxyWindow,ihsWindow,xpehhWindow}   (needs hipcc, no GPU)
#   make oracle     the test-only CPU restatement (+ the unmodified reference tools where /root/reference exists)
#   make test       CPU test suite          make gputest   GPU parity suite (needs an MI355X)
HIPCC    ?= hipcc
ARCH     ?= gfx950
PKG      := popgenomicstools_amd
CSRC     := $(PKG)/csrc
HOST     := $(PKG)/host
BIN      := $(PKG)/bin
LIB      := $(PKG)/libpgtwin.so
LIBSRC   := $(CSRC)/pgt_kernels.hip $(CSRC)/pgt_af_kernels.hip $(CSRC)/pgt_ingest.hip $(CSRC)/pgt_api.cpp $(CSRC)/pgt_windows.cpp
LIBHDR   := $(CSRC)/pgt_internal.h $(CSRC)/pgt_device.h include/pgtwin.h
LIBFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -shared
TOOLS    := fstWindow hetWindow dxyWindow ihsWindow xpehhWindow

all: $(LIB) $(addprefix $(BIN)/,$(TOOLS))

$(LIB): $(LIBSRC) $(LIBHDR)
	$(HIPCC) $(LIBFLAGS) -Iinclude -I$(CSRC) -o $@ $(LIBSRC)

$(BIN)/%: $(HOST)/%_main.cpp $(HOST)/host_common.h $(HOST)/extreme_common.h include/pgtwin.h $(LIB)
	@mkdir -p $(BIN)
	$(HIPCC) -O2 -std=c++17 -Iinclude -I$(HOST) $< -o $@ -L$(PKG) -lpgtwin -lz -lpthread -Wl,-rpath,'$$ORIGIN/..'

oracle:
	$(MAKE) -C oracle all

test: all oracle
	python -m pytest tests -q -m "not gpu"

gputest: all oracle
	python -m pytest tests -q -m gpu

clean:
	rm -f $(LIB) $(PKG)/libpgtwin.flags $(PKG)/libpgtwin.so.lock $(addprefix $(BIN)/,$(TOOLS))

.PHONY: all oracle test gputest clean
