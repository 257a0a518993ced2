// placeholder until the encoder lands (next commit)
#include "vm_internal.h"
struct vm_encoder { vm_ctx *ctx; };
extern "C" int vm_encoder_create(vm_ctx *ctx, const vm_encoder_desc *, const void *const *, int, vm_encoder **) {
    return vm_fail(ctx, VM_ERR_UNSUPPORTED, "encoder not built yet");
}
extern "C" void vm_encoder_destroy(vm_encoder *) {}
extern "C" int vm_encoder_tokens(const vm_encoder *) { return 0; }
extern "C" int vm_encoder_patch_k(const vm_encoder *) { return 0; }
extern "C" int vm_encoder_out_dim(const vm_encoder *) { return 0; }
extern "C" size_t vm_encode_workspace_bytes(const vm_encoder *, int) { return 0; }
extern "C" int vm_encode(vm_encoder *, const void *, int, void *, int, void *, size_t, void *) { return VM_ERR_UNSUPPORTED; }
