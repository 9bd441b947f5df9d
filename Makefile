# Builds the MI355X (gfx950) engine in-tree:
#   whisper.tflite_amd/lib/libwhisper-tflite.so   C ABI (include/wt_capi.h) + C++ surface
#   whisper.tflite_amd/bin/encdec, bin/minimal    the reference's CLIs, same arguments
#   whisper.tflite_amd/bin/wt-make-assets         synthetic weights / vocab writer
#   oracle/libwt_oracle.so (+ oracle/_ref/)       test infrastructure (see oracle/)
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
PKG   := whisper.tflite_amd
SRC   := $(PKG)/csrc
OBJ   := $(PKG)/build
CXXFLAGS := -std=c++17 -O3 -fPIC -Iinclude -I$(SRC) -Wall -Wno-unused-result
HIPFLAGS := $(CXXFLAGS) --offload-arch=$(ARCH) -ffp-contract=fast $(HIPFLAGS_EXTRA)

KERNELS := k_gemm k_gemm_planes k_gemm_bf16 k_misc k_attention k_attention_planes k_decoder k_cross_absorbed
HOSTSRC := engine capi host_util weights_gen whisper_api tflite_extract
OBJS := $(addprefix $(OBJ)/,$(addsuffix .o,$(KERNELS) $(HOSTSRC)))

all: lib apps oracle
lib: $(PKG)/lib/libwhisper-tflite.so
apps: $(PKG)/bin/encdec $(PKG)/bin/minimal $(PKG)/bin/wt-make-assets
oracle:
	$(MAKE) -C oracle

# the attention's softmax runs beside other wavefronts' MFMAs, where packed fp32 instructions (what the SLP vectoriser makes
# of its row sums) cost more than the scalar ones they replace (MI355X_MICROARCH.md, vector-instruction issue costs): 325.8
# against 334.2 us per layer (tools/ab_attn.sh)
$(OBJ)/k_attention_planes.o: HIPFLAGS += -fno-slp-vectorize

$(OBJ)/%.o: $(SRC)/%.hip $(SRC)/kernels.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJ)/%.o: $(SRC)/%.cpp $(wildcard $(SRC)/*.h) $(wildcard include/*.h)
	@mkdir -p $(OBJ)
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(PKG)/lib/libwhisper-tflite.so: $(OBJS)
	@mkdir -p $(PKG)/lib
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS)

$(PKG)/bin/encdec: $(PKG)/app/encdec.cpp $(PKG)/lib/libwhisper-tflite.so
	@mkdir -p $(PKG)/bin
	$(HIPCC) $(CXXFLAGS) -o $@ $< -L$(PKG)/lib -lwhisper-tflite -Wl,-rpath,'$$ORIGIN/../lib'

$(PKG)/bin/minimal: $(PKG)/app/minimal.cpp $(PKG)/lib/libwhisper-tflite.so
	@mkdir -p $(PKG)/bin
	$(HIPCC) $(CXXFLAGS) -o $@ $< -L$(PKG)/lib -lwhisper-tflite -Wl,-rpath,'$$ORIGIN/../lib'

$(PKG)/bin/wt-make-assets: $(PKG)/app/make_assets.cpp $(PKG)/lib/libwhisper-tflite.so
	@mkdir -p $(PKG)/bin
	$(HIPCC) $(CXXFLAGS) -o $@ $< -L$(PKG)/lib -lwhisper-tflite -Wl,-rpath,'$$ORIGIN/../lib'

clean:
	rm -rf $(OBJ) $(PKG)/lib $(PKG)/bin
	$(MAKE) -C oracle clean
.PHONY: all lib apps oracle clean
