"""The drop-in boundary type-checks against the reference's OWN call sites: the reference's C sources are run
through `gcc -std=c99 -fsyntax-only -Werror=incompatible-pointer-types` with this repo's include/ceed.h as <ceed.h> and a TEST-ONLY
stand-in for the PETSc headers (tests/petsc_stub: type names and error macros, no PETSc function declared -- nothing is compiled
to code or linked).  Every diagnostic that names a Ceed* / CEED_* identifier is a hole in the boundary: an undeclared libCEED
function or constant, a wrong argument type, a wrong arity.  Calls to PETSc functions are implicit declarations and are ignored.
Skipped where /root/reference is absent (the GPU box)."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference"
SOURCES = ["src/setuplibceed.c", "src/matops.c", "src/misc.c", "src/boundary.c", "src/cloptions.c", "src/setupdm.c", "elasticity.c"]


@pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("gcc") is None, reason="needs the reference tree and gcc")
@pytest.mark.parametrize("src", SOURCES)
def test_reference_sources_type_check_against_the_boundary(src):
    cmd = ["gcc", "-std=c99", "-fsyntax-only", "-fno-diagnostics-show-caret", "-Werror=incompatible-pointer-types", "-Werror=int-conversion",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "petsc_stub"), os.path.join(REF, src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    diags = [l for l in r.stderr.splitlines() if re.search(r": (error|warning):", l)]
    implicit = [re.search(r"implicit declaration of function '([^']+)'", l) for l in diags]
    implicit_names = {m.group(1) for m in implicit if m}
    # a libCEED entry point the reference calls but include/ceed.h does not declare
    assert not [n for n in implicit_names if n.startswith(("Ceed", "CEED_"))], sorted(implicit_names)
    # everything else: no diagnostic may mention a Ceed* / CEED_* identifier (the "did you mean 'CeedX'" hints gcc attaches to
    # an unknown PETSc name are not about the boundary)
    others = [l for l in diags if "implicit declaration of function" not in l]
    others = [re.sub(r"; did you mean '[^']+'\?", "", l) for l in others]
    bad = [l for l in others if re.search(r"\b(Ceed[A-Za-z0-9_]*|CEED_[A-Z0-9_]+)\b", l)]
    assert not bad, "\n".join(bad)
    # and no hard error at all: the PETSc stand-in is complete enough that what is left are implicit declarations
    errors = [l for l in others if ": error:" in l]
    assert not errors, "\n".join(errors[:20])


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree")
def test_every_ceed_identifier_the_reference_uses_is_declared():
    """Token-level cross-check: each Ceed* / CEED_* identifier in the reference's sources is declared by include/ceed.h, or is one
    of the reference's own names (its CeedData struct family, the CEED_DIR Makefile variable quoted in a comment)."""
    hdr = open(os.path.join(ROOT, "include", "ceed.h")).read()
    declared = set(re.findall(r"\b(Ceed[A-Za-z0-9_]*|CEED_[A-Z0-9_]+)\b", hdr))
    own = set()
    used = set()
    for src in SOURCES + ["elasticity.h"]:
        txt = open(os.path.join(REF, src)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        txt = re.sub(r"//[^\n]*", "", txt)
        txt = re.sub(r'"(?:\\.|[^"\\])*"', '""', txt)                       # string literals (option prefixes such as "CEED_MEM_")
        used |= set(re.findall(r"\b(Ceed[A-Za-z0-9_]*|CEED_[A-Z0-9_]+)\b", txt))
        own |= set(re.findall(r"(?:struct|typedef struct)\s+(Ceed[A-Za-z0-9_]*)", txt))
        own |= set(re.findall(r"}\s*\*?\s*(Ceed[A-Za-z0-9_]*)\s*;", txt))
        own |= set(re.findall(r"typedef\s+struct\s+\w+\s*\*\s*(Ceed[A-Za-z0-9_]*)\s*;", txt))
        own |= set(re.findall(r"PetscErrorCode\s+(Ceed[A-Za-z0-9_]*)\s*\(", txt))       # functions the reference defines itself
    missing = sorted(used - declared - own)
    assert not missing, missing
