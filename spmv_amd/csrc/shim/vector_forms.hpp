// shim/vector_forms.hpp -- form selectors of the CSR-vector schedule, shared by both translation units
#pragma once
// Kernel forms of the CSR-vector schedule.  Which one is fastest differs between MI355X boxes by a
// few percent, so create() times the applicable ones once on the resident matrix
// (autotune_vector) and keeps the winner in d->vec_choice; option vector_form forces one (tests, A/B runs).
enum { VEC_AUTO = 0, VEC_PIPE = 4, VEC_TILE_D2 = 5, VEC_TILE_D8 = 6, VEC_TILE_D4 = 10, VEC_TILE_D4_NOPRE = 11, VEC_TILE_D2_NOPRE = 12 };

