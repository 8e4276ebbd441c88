// faiss_amd/csrc/selector.cpp -- host side of faiss::IDSelector (faiss/impl/IDSelector.h:21-215, IDSelector.cpp):
// membership tests with the reference's semantics, and the translation of a selector tree into the postfix program the
// device evaluates (kernels.h SelProgram, selector_kernels.hip).
#include <algorithm>
#include "index.h"
#include "kernels.h"

namespace faiss_amd {

static SelInstr& sel_push(SelProgram& prog, int op) {
    FA_THROW_IF_NOT_MSG(prog.n < kSelMaxInstr, "IDSelector expression too large for the device program");
    SelInstr& in = prog.ins[prog.n++];
    in.op = op;
    in.a = in.b = 0;
    in.ptr = nullptr;
    return in;
}

// device copy of a host array, made once per device (selectors are immutable after construction; the current device is
// `device`, set by the search that compiles the selector)
static const void* sel_upload(std::mutex& mu, std::map<int, DevBuf>& copies, const void* src, size_t bytes, int device,
                              hipStream_t stream) {
    std::lock_guard<std::mutex> g(mu);
    DevBuf& dev = copies[device];
    if (!dev.p) {
        dev.ensure(std::max<size_t>(bytes, 16));
        if (bytes) HIP_CHECK(hipMemcpyAsync(dev.p, src, bytes, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream)); // the host array may be pageable
    }
    return dev.p;
}

void IDSelectorAll::compile(SelProgram& prog, int, hipStream_t) const {
    sel_push(prog, SEL_ALL);
}

void IDSelectorRange::compile(SelProgram& prog, int, hipStream_t) const {
    SelInstr& in = sel_push(prog, SEL_RANGE);
    in.a = imin;
    in.b = imax;
}

IDSelectorBatch::IDSelectorBatch(size_t n, const idx_t* indices) : ids(indices, indices + n) {
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
}
bool IDSelectorBatch::is_member(idx_t id) const {
    return std::binary_search(ids.begin(), ids.end(), id);
}
void IDSelectorBatch::compile(SelProgram& prog, int device, hipStream_t stream) const {
    SelInstr& in = sel_push(prog, SEL_SET);
    in.a = (int64_t)ids.size();
    in.ptr = sel_upload(mu_, dev_, ids.data(), ids.size() * sizeof(idx_t), device, stream);
}

void IDSelectorBitmap::compile(SelProgram& prog, int device, hipStream_t stream) const {
    SelInstr& in = sel_push(prog, SEL_BITMAP);
    in.a = (int64_t)bitmap.size();
    in.ptr = sel_upload(mu_, dev_, bitmap.data(), bitmap.size(), device, stream);
}

void IDSelectorNot::compile(SelProgram& prog, int device, hipStream_t stream) const {
    FA_THROW_IF_NOT_MSG(sel, "IDSelectorNot without an operand");
    sel->compile(prog, device, stream);
    sel_push(prog, SEL_NOT);
}

bool IDSelectorBinary::is_member(idx_t id) const {
    const bool a = lhs->is_member(id), b = rhs->is_member(id);
    return op == SEL_AND ? (a && b) : op == SEL_OR ? (a || b) : (a != b);
}
void IDSelectorBinary::compile(SelProgram& prog, int device, hipStream_t stream) const {
    FA_THROW_IF_NOT_MSG(lhs && rhs && (op == SEL_AND || op == SEL_OR || op == SEL_XOR), "malformed binary IDSelector");
    lhs->compile(prog, device, stream);
    rhs->compile(prog, device, stream);
    sel_push(prog, op);
}

} // namespace faiss_amd
