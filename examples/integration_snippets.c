/* Every cap_* call INTEGRATION.md section B shows, as ONE C translation unit (gcc -std=c99, conversions between integers and
 * pointers and incompatible pointer types are errors): tests/test_abi.py compiles this file and checks that each statement of the
 * document's code blocks appears here verbatim - round 4's document had four calls that did not match the header (wrong arity, characters
 * for enums, a wrong signature) and nothing noticed.  The function is never called (the drivers next to this file are the runnable
 * forms); generated from the document, keep the two in step.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "capital_amd.h"

int integration_md_section_b(void) {
  int st = 0;
  /* the names the document uses */
  const int64_t N = 4096, M = 4096, K = 4096, nb = 512, n = 256, m_local = 1 << 20, cols = 4096, rows = 4096, gx = 1, gy = 1;
  const int64_t split = 1, bcMultiplier = -2, ld = 4096, lda = 4096, ldb = 4096, ldc = 4096, ldt = 4096, ldbc = 4096, ldp = 128, ld_host = 4096, count = 1 << 20;
  const int complete_inv = 1, rank = 0, size = 1, c = 1, num_chunks = 0, Pr = 1, Pc = 1, pr = 0, pc = 0, x = 0, y = 0, d = 1;
  const int64_t key = 0;
  const double alpha = 1.0, beta = 0.0;
  void* stream = NULL;
  unsigned char id128[128];
  double *A = NULL, *R = NULL, *Rinv = NULL, *Alocal = NULL, *Rlocal = NULL, *Apiece = NULL, *Rpiece = NULL, *RinvPiece = NULL, *bc = NULL;
  double *host_A = NULL, *host_R = NULL, *host_piece = NULL, *device_ptr = NULL, *Aloc = NULL, *Bloc = NULL, *Cloc = NULL, *Tpiece = NULL, *Bpiece = NULL;
  double *Cpiece = NULL, *tmp = NULL, *Q = NULL;
  int64_t info = 0, lc = 0, piece = 0, ldq = 0, ldr = 0;
  cap_cholinv_plan* pack = NULL; cap_comm* world = NULL; cap_topo *grid = NULL, *rect = NULL; cap_redist_plan* rp = NULL;
  cap_desc *dA = NULL, *dR = NULL, *dA1 = NULL, *dR1 = NULL, *dm = NULL, *dv = NULL; cap_dist2d_plan* p2 = NULL; cap_summa_plan* sp = NULL;
  cap_cacqr_plan *qp = NULL, *qp3 = NULL;
  st |= cap_fill_symmetric(A, N, N, /*x*/0, /*y*/0, /*d*/1, /*diagonallyDominant*/1, NULL);
  st |= cap_cholinv_plan_create(&pack, N, complete_inv, split, bcMultiplier, 'U', NULL);
  st |= cap_cholinv_factor(pack, A, N, NULL);
  st |= cap_cholinv_info(pack, NULL, &info);
  st |= cap_cholinv_get_R(pack, R, N, NULL);
  st |= cap_cholinv_get_Rinv(pack, Rinv, N, NULL);
  st |= cap_comm_unique_id(id128);
  st |= cap_comm_create(&world, id128, rank, size, NULL);
  st |= cap_topo_create(&grid, /*square*/0, world, c, /*layout*/0, num_chunks);
  st |= cap_cholinv_plan_create(&pack, N, complete_inv, split, bcMultiplier, 'U', world);
  lc = cap_bc_num_local_cols(N, 512, size, rank);
  st |= cap_fill_symmetric_bc(Alocal, N, N, 512, size, rank, 1, stream);
  st |= cap_cholinv_factor(pack, Alocal, N, stream);
  st |= cap_cholinv_info(pack, stream, &info);
  st |= cap_cholinv_get_R(pack, Rlocal, N, stream);
  st |= cap_cholinv_set_option(pack, "cyclic_c", c);
  piece = cap_cholinv_get_option(pack, "piece");
  st |= cap_fill_symmetric(Apiece, piece, N, x, y, d, 1, stream);
  st |= cap_cholinv_factor(pack, Apiece, piece, stream);
  st |= cap_cholinv_get_R(pack, Rpiece, piece, stream);
  st |= cap_cholinv_get_Rinv(pack, RinvPiece, piece, stream);
  st |= cap_redist_plan_create(&rp, N, nb, world, c, Pr);
  st |= cap_redistribute_cyclic_to_bc(rp, Apiece, piece, bc, ldbc, stream);
  st |= cap_redistribute_bc_to_cyclic(rp, bc, ldbc, Rpiece, piece, stream);
  st |= cap_desc_create_bc(&dA, N, N, nb, Pr, Pc, pr, pc, NULL, 0);
  st |= cap_desc_create_bc(&dR, N, N, nb, Pr, Pc, pr, pc, NULL, 0);
  st |= cap_desc_import_host_global(dA, host_A, N, stream);
  st |= cap_dist2d_plan_create(&p2, N, nb, world, Pr, NULL, NULL);
  st |= cap_dist2d_factor_desc(p2, dA, stream);
  st |= cap_dist2d_info(p2, stream, &info);
  st |= cap_dist2d_get_R_desc(p2, dR, stream);
  st |= cap_desc_export_host_global(dR, host_R, N, stream);
  st |= cap_desc_create_bc(&dA1, N, N, 512, 1, size, 0, rank, NULL, 0);
  st |= cap_desc_create_bc(&dR1, N, N, 512, 1, size, 0, rank, NULL, 0);
  st |= cap_cholinv_factor_desc(pack, dA1, stream);
  st |= cap_cholinv_get_R_desc(pack, dR1, stream);
  st |= cap_summa_plan_create(&sp, grid, M, N, K, num_chunks);
  st |= cap_fill_random(Aloc, lda, M, K, x, y, d, d, key, stream);
  st |= cap_summa_dgemm(sp, alpha, Aloc, lda, Bloc, ldb, beta, Cloc, ldc, stream);
  st |= cap_summa_dtrmm(sp, CAP_LEFT, CAP_UPPER, CAP_TRANS, CAP_NONUNIT, alpha, Tpiece, ldt, /*t_packed*/1, Bpiece, ldb, stream);
  st |= cap_summa_dsyrk(sp, CAP_UPPER, CAP_TRANS, alpha, Apiece, lda, beta, Cpiece, ldc, /*c_packed*/0, stream);
  st |= cap_util_transpose(grid, Tpiece, tmp, count, stream);
  st |= cap_cacqr_plan_create(&qp, m_local, n, 2, world);
  st |= cap_topo_create(&rect, /*rect*/1, world, c, 0, 0);
  st |= cap_cacqr_plan_create_grid(&qp3, M, N, 2, rect);
  st |= cap_cacqr_factor(qp, Alocal, lda, stream);
  Q = cap_cacqr_Q_ptr(qp, &ldq);
  R = cap_cacqr_R_ptr(qp, &ldr);
  st |= cap_cacqr_R_piece(qp3, Rpiece, ldp, stream);
  st |= cap_desc_create(&dm, cols, rows, gx, gy);
  st |= cap_desc_create_view(&dv, cols, rows, gx, gy, device_ptr, ld);
  st |= cap_desc_import_host(dm, host_piece, ld_host, stream);
  st |= cap_desc_export_host(dm, host_piece, ld_host, stream);
  (void)lc; (void)piece; (void)Q; (void)R; (void)info;
  return st;
}

int main(int argc, char** argv) {
  (void)argv;
  if (argc > 1000) return integration_md_section_b();      /* compiled and linked, never executed */
  printf("integration snippets: compiled against capital_amd.h\n");
  return 0;
}
