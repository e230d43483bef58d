/*
 * ref_qfunctions_wrap.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Export table for the REFERENCE's own QFunctions.  Nothing from the reference
 * is copied: its qfunctions/ *.h are #included where they lie under
 * /root/reference (-I on the compile line, see oracle/Makefile) and compiled
 * against this repo's boundary header include/ceed.h -- the same header the
 * reference's translation units would use in a drop-in build -- into
 * oracle/_ref/libref_qfunctions.so (git-ignored; travels to the GPU box).
 * The reference's callbacks are `static`, so this table is the only way out.
 */
#include <ceed.h>
#include <string.h>

#include "common.h"
#include "linElas.h"
#include "hyperSS.h"
#include "hyperFS.h"
#include "constantForce.h"
#include "manufacturedForce.h"
#include "manufacturedTrue.h"

#define REF_ENTRY(n) {#n, n, n##_loc}
static const struct { const char *name; CeedQFunctionUser f; const char *loc; } ref_tab[] = {
    REF_ENTRY(SetupGeo),   REF_ENTRY(LinElasF),  REF_ENTRY(LinElasdF),
    REF_ENTRY(HyperSSF),   REF_ENTRY(HyperSSdF), REF_ENTRY(HyperFSF),
    REF_ENTRY(HyperFSdF),  REF_ENTRY(SetupConstantForce),
    REF_ENTRY(SetupMMSForce), REF_ENTRY(MMSTrueSoln),
    REF_ENTRY(LinElasEnergy), REF_ENTRY(HyperSSEnergy), REF_ENTRY(HyperFSEnergy),
    REF_ENTRY(LinElasDiagnostic), REF_ENTRY(HyperSSDiagnostic), REF_ENTRY(HyperFSDiagnostic),
};

CEED_EXTERN CeedQFunctionUser RefGetQFunction(const char *name) {
  for (size_t i = 0; i < sizeof ref_tab / sizeof ref_tab[0]; i++)
    if (!strcmp(ref_tab[i].name, name)) return ref_tab[i].f;
  return NULL;
}
CEED_EXTERN const char *RefGetQFunctionLoc(const char *name) {
  for (size_t i = 0; i < sizeof ref_tab / sizeof ref_tab[0]; i++)
    if (!strcmp(ref_tab[i].name, name)) return ref_tab[i].loc;
  return NULL;
}
