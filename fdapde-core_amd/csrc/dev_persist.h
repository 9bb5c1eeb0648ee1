// dev_persist.h -- the persistent CG's resident layout built on the device (dev_persist.hip); plain C++ interface used by capi.hip.
#ifndef FDAPDE_DEV_PERSIST_H
#define FDAPDE_DEV_PERSIST_H

#include <cstdint>
#include <string>

#include "internal.h"

namespace fdapde_hip {

struct DevPersist {   // device arrays (hipMalloc'd here; ownership passes to the caller), same contents as PersistLayout's vectors
    int32_t *slot_dof = nullptr, *sl_off = nullptr, *ell_src = nullptr, *exp_off = nullptr, *imp_off = nullptr, *imp_pos = nullptr;
    int64_t* ell_off = nullptr;
    uint16_t *ell_code = nullptr, *exp_slot = nullptr;
    int32_t* drop_dof = nullptr;   // blocked mode: the rows the layout leaves out (Dirichlet DOFs), ascending; pl.n_drop of them
};
void dev_persist_release(DevPersist* p);
// rowptr / colidx / bnd: the internal pattern and boundary flags on the device.  Fills the sizes of `pl` (G, R, nsl, n_int, n_entries,
// nnz, n_board, max_imp, max_exp, max_block); its vectors stay empty.  FDAPDE_EUNSUPPORTED: the system does not qualify (as the host
// builder) or a row is longer than 255 entries (the caller then asks the host builder).
// blocked_rows > 0: the layout of the blocked-ELL SpMV instead (k_spmv_blocked): workgroups of about that many rows, as many as the
// system needs; imp_pos holds DOF ids of the global vector (ascending inside a workgroup); no exports / board.
// block_rows != nullptr (persistent mode only): exactly n_wg workgroups, workgroup g owning block_rows[g] >= 1 consecutive interior rows
// (they add up to the interior row count) instead of equal shares -- the speed-weighted split of capi.hip's calibration.
// balance (persistent mode, no block_rows): workgroup boundaries at equal cost (stored entries + 2 per row) instead of equal row counts.
// sym_mode: symmetric storage (internal.h persist_sym_owner / persist_want_sym; persistent mode only); pl.sym says what was built.
int dev_build_persist_layout(int64_t nd, int32_t max_row, const int32_t* d_rowptr, const int32_t* d_colidx, const uint8_t* d_bnd, bool use_bnd,
                             int n_wg, int lds_entries, int blocked_rows, const int32_t* block_rows, int sym_mode, bool balance, void* stream, PersistLayout& pl, DevPersist* out,
                             std::string& err);

void dev_persist_preload();    // loads this unit's code object
}  // namespace fdapde_hip
#endif
