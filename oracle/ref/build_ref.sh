#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY — builds oracle/_ref/librvb_ref.so from the reference's own
# kernel source where it lies under /root/reference.
#
# * The OpenCL C text is the raw string in reference rayverb/kernel.cpp:9-627; the two
#   #defines the reference pastes in front of it (rayverb/kernel.cpp:7-8, values from
#   rayverb/clstructs.h:4-5 through std::to_string) are re-created here.
# * The text is extracted into a temporary directory that is deleted on exit; only the
#   shared object lands in oracle/_ref/ (git-ignored, but shipped to the GPU box).
# * Runs only where /root/reference exists (the build container).  Exit code 3 = skipped.
set -euo pipefail

REF_ROOT="${RVB_REFERENCE_ROOT:-/root/reference}"
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../_ref"
CLANG="${RVB_CLANG:-/opt/rocm/lib/llvm/bin/clang}"
SRC="$REF_ROOT/rayverb/kernel.cpp"

if [ ! -f "$SRC" ]; then
    echo "build_ref: $SRC not present, skipping (prebuilt oracle/_ref is used if it exists)" >&2
    exit 3
fi

TMP="$(mktemp -d /tmp/rvb_ref_build.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$OUT"

# Everything between the opening R"( and the closing )" of the raw string literal.
python3 - "$SRC" "$TMP/rvb_ref_kernel_text.cl" <<'PY'
import sys
src = open(sys.argv[1]).read()
begin = src.index('R"(') + 3
end = src.rindex(')"')
with open(sys.argv[2], 'w') as f:
    f.write('#define NUM_IMAGE_SOURCE 10\n')
    f.write('#define SPEED_OF_SOUND 340.000000\n')
    f.write(src[begin:end])
PY

# `build_ref.sh alt`: a second library, librvb_ref_alt.so, with the alternative conforming built-ins of ref_builtins.cl and the
# kernel text contracted into fused multiply-adds (FP_CONTRACT ON, the OpenCL default) — input of oracle/sensitivity.py only.
VARIANT="${1:-}"
CONTRACT=off; EXTRA=(); NAME=librvb_ref.so
if [ "$VARIANT" = "alt" ]; then CONTRACT=fast; EXTRA=(-DRVB_REF_ALT_BUILTINS -mfma); NAME=librvb_ref_alt.so; fi
CLFLAGS=(-x cl -cl-std=CL1.2 -Xclang -finclude-default-header
         -target x86_64-unknown-linux-gnu -mavx2 -ffp-contract=$CONTRACT -O1 -w -fPIC -I "$TMP" "${EXTRA[@]}")

"$CLANG" "${CLFLAGS[@]}" -c "$HERE/ref_wrap.cl" -o "$TMP/ref_wrap.o"
"$CLANG" "${CLFLAGS[@]}" -c "$HERE/ref_builtins.cl" -o "$TMP/ref_builtins.o"
gcc -O2 -fPIC -fopenmp -ffp-contract=off -c "$HERE/ref_harness.c" -o "$TMP/ref_harness.o"
gcc -shared -fopenmp -o "$OUT/$NAME" "$TMP/ref_harness.o" "$TMP/ref_wrap.o" "$TMP/ref_builtins.o" -lm

echo "build_ref: wrote $OUT/$NAME"
