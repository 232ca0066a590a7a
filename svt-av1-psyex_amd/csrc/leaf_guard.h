// leaf_guard.h -- how the pointer-level entries (include/svt_hip_leaf.h) fail: CLOSED.
//
// The reference's dispatch always leaves a working kernel in every function pointer (Codec/aom_dsp_rtcd.c:38-48,73-99: SET_* picks the
// best variant the CPU allows, the `_c` body at worst) and its leaf kernels have no error channel.  A `_hip` entry that finds no bound
// context, or whose device work fails, therefore hands the call -- same arguments -- to the kernel that sat in the encoder's pointer before
// svt_hip_install_rtcd overwrote it (the encoder's own: never the oracle), and the failure is kept for svt_hip_leaf_status / svt_hip_last_error.
// Mechanics: the helpers throw LeafFailure, every exported entry is a function-try-block whose handler (LEAF_CATCH) resolves the previous
// kernel by the entry's own symbol name.  An entry nobody installed over (a direct call, no previous kernel) that is nested inside another
// entry passes the failure on; at the outermost level it records the error and returns zero / leaves its outputs alone.
#ifndef SVT_HIP_LEAF_GUARD_H
#define SVT_HIP_LEAF_GUARD_H
#include <hip/hip_runtime_api.h>
#include <mutex>
#include "svt_hip_internal.h"

struct LeafFailure { char what[200]; };

[[noreturn]] void leaf_fail(const char *fmt, ...);                 // formats the message and throws LeafFailure
SvtHipContext *leaf_ctx();                                          // the bound context; throws when there is none (or a failure is injected)
void           leaf_check(SvtHipContext *ctx, hipError_t e, const char *what); // throws on a HIP error
uint8_t       *leaf_scratch(SvtHipContext *ctx, size_t bytes);      // lane 0's staging area, grown on demand; throws when out of memory
std::mutex    &leaf_mutex();                                        // the pointer-level entries run one at a time on the context stream
int            leaf_depth();
const void    *leaf_previous(const char *symbol);                   // the kernel svt_hip_install_rtcd found in the slot it gave to `symbol`
void           leaf_note_fallback(const char *symbol, const LeafFailure &f);
void           leaf_note_unhandled(const char *symbol, const LeafFailure &f);

struct LeafEnter { // nesting depth of pointer-level entries on this thread (entries call each other: the facade, the per-size variances)
    LeafEnter();
    ~LeafEnter();
};

template <class R, class... A> R leaf_zero(A...) { return R(); }

template <class R, class... A> auto leaf_resolve(R (*)(A...), const char *symbol, const LeafFailure &f) -> R (*)(A...) {
    if (const void *prev = leaf_previous(symbol)) {
        leaf_note_fallback(symbol, f);
        return reinterpret_cast<R (*)(A...)>(const_cast<void *>(prev));
    }
    if (leaf_depth() > 0) throw f; // an enclosing entry (which may have a previous kernel) decides
    leaf_note_unhandled(symbol, f);
    return &leaf_zero<R, A...>;
}

// RET name(params) LEAF_TRY ...body... LEAF_CATCH(name, args...)
#define LEAF_TRY try { LeafEnter leaf_enter_;
#define LEAF_CATCH(FN, ...) } catch (const LeafFailure &lf_) { return leaf_resolve(&FN, #FN, lf_)(__VA_ARGS__); }

#endif
