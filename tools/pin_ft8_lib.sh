#!/bin/bash
# pin_ft8_lib.sh <path-to-a-kgoba/ft8_lib-checkout> -- pins the CPU oracle to the REAL ft8_lib, the day somebody has it.
#
# The reference calls ft8_find_sync / ft8_decode / pack77 / ft8_encode of kgoba/ft8_lib (rtlsdr_ft8d.c:1450, :1476, :927,
# :934), a git submodule that is EMPTY in the reference snapshot this repository was built from (.gitmodules:1-3; the
# Makefile links its objects, :9).  The oracle restates those functions from the published algorithm; the only vectors of
# the reference itself that reach them are one encode KAT and one +30 dB decode ("parity unpinned").
# This script turns that into one command: it compiles upstream's own ft8/*.c with gcc, runs tools/pin_ft8_lib/pin_harness.c
# against them and against the oracle on the same waterfalls, and diffs
#     ordered candidate lists at caps 7 / 120 / 480, every ft8_decode outcome (ok, ldpc_errors, CRCs, unpack status, hash,
#     text), all 35 856 sync scores, the normalised LLRs (bit patterns), the iteration belief propagation converges in,
#     the decoded bits, and pack77 / ft8_encode of a message list.
# Exit 0 and "PINNED" = the oracle (and with it every GPU parity claim of this repository) is tied to upstream's code for
# the revision given.  A difference is printed as a unified diff with the record that differs.
#
# Needs: gcc, python3 + numpy, this repository (oracle/ builds by itself).  No GPU.
set -euo pipefail
UP=${1:?usage: tools/pin_ft8_lib.sh <path-to-kgoba/ft8_lib>}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
P=$ROOT/tools/pin_ft8_lib
W=${PIN_WORKDIR:-$(mktemp -d /tmp/pin_ft8_lib.XXXXXX)}
[ -f "$UP/ft8/decode.c" ] || { echo "$UP/ft8/decode.c not found: not an ft8_lib checkout"; exit 2; }
make -s -C "$ROOT/oracle"
python3 "$P/make_inputs.py" "$W" ${PIN_LIGHT:+--light}
CFLAGS="-O2 -std=gnu17 -ffp-contract=off -fno-fast-math"      # one IEEE operation per float operation on both sides
# (a) upstream: its own sources; decode.c comes in through upstream_internals.c (which wraps its file-local functions)
UPSRC=""
for f in constants crc ldpc unpack text pack encode; do [ -f "$UP/ft8/$f.c" ] && UPSRC="$UPSRC $UP/ft8/$f.c"; done
gcc $CFLAGS -DPIN_INTERNALS -I"$UP" -I"$P" "$P/pin_harness.c" "$P/upstream_internals.c" $UPSRC -lm -o "$W/pin_upstream"
# (b) the oracle under the same names
gcc $CFLAGS -DPIN_INTERNALS -I"$ROOT/include/ft8_lib" -I"$ROOT/oracle" -I"$P" "$P/pin_harness.c" "$P/oracle_as_ft8_lib.c" \
    -L"$ROOT/oracle" -lft8oracle -Wl,-rpath,"$ROOT/oracle" -lm -o "$W/pin_oracle"
"$W/pin_upstream" "$W/waterfalls.bin" "$W/messages.txt" > "$W/upstream.txt"
"$W/pin_oracle" "$W/waterfalls.bin" "$W/messages.txt" > "$W/oracle.txt"
echo "upstream revision: $(git -C "$UP" rev-parse HEAD 2>/dev/null || echo unknown)"
echo "records: $(wc -l < "$W/oracle.txt") (oracle), $(wc -l < "$W/upstream.txt") (upstream); dumps in $W"
if diff -u "$W/upstream.txt" "$W/oracle.txt" > "$W/pin.diff"; then
    echo "PINNED: the oracle and upstream ft8_lib agree on every record"
else
    head -60 "$W/pin.diff"
    echo "DIFFERENT: $(grep -c '^[-+][^-+]' "$W/pin.diff") lines differ (full diff: $W/pin.diff)"
    exit 1
fi
