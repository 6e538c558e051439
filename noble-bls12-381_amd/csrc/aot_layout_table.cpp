// aot_layout_table.cpp -- the generated slot placements (aot_layout.inc, written by aot_gen at build time; aot_layout.h)
#include "aot_layout.h"
namespace nbls {
namespace {
#define AOT_LAYOUT_POS(NAME, ...) const uint16_t layout_pos_##NAME[] = {__VA_ARGS__};
#define AOT_LAYOUT_ROW(NAME, G, SLOTS, NSTEPS, HASH)
#include "aot_layout.inc"
#undef AOT_LAYOUT_POS
#undef AOT_LAYOUT_ROW
#define AOT_LAYOUT_POS(NAME, ...)
#define AOT_LAYOUT_ROW(NAME, G, SLOTS, NSTEPS, HASH) {#NAME, G, SLOTS, NSTEPS, HASH, layout_pos_##NAME},
const AotLayoutEntry table[] = {
#include "aot_layout.inc"
  {nullptr, 0, 0, 0, 0, nullptr}};
}  // namespace
const AotLayoutEntry* aot_layout_table(size_t* n) { *n = sizeof(table) / sizeof(table[0]) - 1; return table; }
}  // namespace nbls
