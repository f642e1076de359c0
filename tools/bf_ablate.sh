#!/bin/bash
# k_bf2 with parts of its work compiled out (timing experiment, wrong results by construction).  Variant libraries are built
# on the build host with tools/buildvar.sh, e.g.
#   tools/buildvar.sh nostore -DBF2_NOSTORE      tools/buildvar.sh nopass -DBF2_NOPASS      tools/buildvar.sh nh1 -DBF2_NH=1
# and travel to the GPU box with the snapshot; this script times them (tools/var_try.sh does the same for any list).
cd "$GRAFT_REPO_ROOT"
bash tools/var_try.sh base nostore nopass nh1
