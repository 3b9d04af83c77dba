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
# The checkout must be of the API era the reference's call sites imply; any other is refused (exit 3) with the revision to
# check out -- see "which API era" below.  PIN_CHECK_ERA_ONLY=1 stops after that check.
#
# Needs: gcc, python3 + numpy, this repository (oracle/ builds by itself).  No GPU.
set -euo pipefail
UP=${1:?usage: tools/pin_ft8_lib.sh <path-to-kgoba/ft8_lib>}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
P=$ROOT/tools/pin_ft8_lib
W=${PIN_WORKDIR:-$(mktemp -d /tmp/pin_ft8_lib.XXXXXX)}
[ -f "$UP/ft8/decode.c" ] && [ -f "$UP/ft8/decode.h" ] || { echo "$UP/ft8/decode.{c,h} not found: not an ft8_lib checkout"; exit 2; }

# ---- which API era is this checkout? ---------------------------------------------------------------------------------
# The reference's call sites fix the era of the submodule commit it was built against (the commit itself is not recorded in
# the snapshot: .gitmodules has the URL only):
#   rtlsdr_ft8d.c:1439-1448  candidate_t, and waterfall_t filled with .num_blocks .num_bins .time_osr .freq_osr .mag
#                            .block_stride .protocol = PROTO_FT8   -> AFTER FT4 support gave waterfall_t `block_stride` and
#                            `ftx_protocol_t protocol`
#   :1450  ft8_find_sync(&power, K_MAX_CANDIDATES, candidate_list, K_MIN_SCORE)
#   :1476  ft8_decode(&power, cand, &message, K_LDPC_ITERS, &status)   with message_t {text, hash}, decode_status_t
#   :927   pack77(message, packed)      :934  ft8_encode(packed, tones)
#                            -> BEFORE the ftx_* rename (ftx_waterfall_t, ftx_find_candidates, ftx_decode_candidate, ft8/message.h)
# A checkout outside that window cannot be compared (other structs, other signatures): it is refused here, with the
# revision to check out, instead of failing somewhere inside gcc.
H="$UP/ft8/decode.h"
has() { grep -Eq "$1" "$2" 2>/dev/null; }
era=ok
if has 'ftx_find_candidates|ftx_decode_candidate|ftx_waterfall_t' "$H"; then era=too-new
elif ! has '\bft8_find_sync[[:space:]]*\(' "$H" || ! has '\bft8_decode[[:space:]]*\(' "$H"; then era=too-old
elif ! has 'ftx_protocol_t[[:space:]]+protocol' "$H" || ! has 'block_stride' "$H"; then era=too-old
elif ! has '\bpack77[[:space:]]*\(' "$UP/ft8/pack.h" || ! has '\bft8_encode[[:space:]]*\(' "$UP/ft8/encode.h" || ! has '\bunpack77[[:space:]]*\(' "$UP/ft8/unpack.h"; then era=too-new
fi
if [ "$era" != ok ]; then
    echo "REFUSED: $UP is an ft8_lib checkout of the wrong API era ($era) for Guenael/rtlsdr-ft8d."
    if [ "$era" = too-new ]; then
        echo "  ft8/decode.h declares the later ftx_* interface (ftx_find_candidates / ftx_decode_candidate / ftx_waterfall_t, messages through"
        echo "  ft8/message.h); the reference calls ft8_find_sync / ft8_decode / pack77 / ft8_encode (rtlsdr_ft8d.c:1450, :1476, :927, :934)."
    else
        echo "  ft8/decode.h lacks ft8_find_sync / ft8_decode, or its waterfall_t has no 'block_stride' / 'ftx_protocol_t protocol' field, which"
        echo "  the reference sets at rtlsdr_ft8d.c:1440-1448: this revision predates FT4 support."
    fi
    echo "  Wanted: any commit AFTER waterfall_t gained 'ftx_protocol_t protocol' (FT4 support) and BEFORE the ftx_* rename of ft8/decode.h."
    echo "  In a full clone of https://github.com/kgoba/ft8_lib these two commands name the window:"
    echo "      git log --reverse --format='%h %ad %s' --date=short -G'ftx_protocol_t[[:space:]]+protocol' -- ft8/decode.h | head -1    # first commit inside"
    echo "      git log --reverse --format='%h %ad %s' --date=short -G'ftx_find_candidates' -- ft8/decode.h | head -1                  # first commit outside (take its parent)"
    if git -C "$UP" rev-parse --git-dir >/dev/null 2>&1; then
        first_in=$(git -C "$UP" log --all --reverse --format=%H -G'ftx_protocol_t[[:space:]]+protocol' -- ft8/decode.h 2>/dev/null | sed -n 1p)      # (sed reads to the end: no SIGPIPE under pipefail)
        first_out=$(git -C "$UP" log --all --reverse --format=%H -G'ftx_find_candidates' -- ft8/decode.h 2>/dev/null | sed -n 1p)
        if [ -n "$first_in" ]; then
            echo "  This clone's history: first commit inside the window  $(git -C "$UP" log -1 --format='%h %ad %s' --date=short "$first_in")"
            if [ -n "$first_out" ] && git -C "$UP" rev-parse -q --verify "$first_out~1" >/dev/null; then
                echo "                        last commit inside the window   $(git -C "$UP" log -1 --format='%h %ad %s' --date=short "$first_out~1")"
                echo "  SUGGESTED: git -C $UP checkout $(git -C "$UP" rev-parse "$first_out~1")    # then run this script again"
            else
                echo "  SUGGESTED: git -C $UP checkout $first_in    # (no ftx_* rename in this history: any later commit that still has ft8_find_sync works too)"
            fi
        else
            echo "  This clone's history has no commit inside the window (shallow clone?): fetch the full history (git fetch --unshallow)."
        fi
    else
        echo "  ($UP is not a git work tree: fetch a full clone to pick the revision.)"
    fi
    echo "  The reference was last touched for this interface in 2021 (rtlsdr_ft8d.c:3); a commit of late 2021 / early 2022 is the likeliest pin."
    exit 3
fi
echo "API era: ft8_find_sync / ft8_decode with waterfall_t.protocol -- the interface rtlsdr_ft8d.c:1439-1494 calls"
[ -n "${PIN_CHECK_ERA_ONLY:-}" ] && exit 0
# ---- how deep can the comparison go with THIS checkout? -----------------------------------------------------------------
# Beyond the public calls the harness compares three stages behind them (every sync score, the normalised LLR bit patterns, BP
# convergence) through upstream's file-local functions, which upstream_internals.c reaches by including ft8/decode.c.  Their
# names are those of the era's revisions as far as the call sites tell; a revision that spells one differently still gets the
# public-interface comparison (ordered candidate lists at three caps, every ft8_decode outcome incl. text / CRCs / hash,
# pack77, ft8_encode) instead of a compiler error, and the note says which -DPIN_UPSTREAM_*=<name> brings the rest back.
INT=-DPIN_INTERNALS
missing=""
for n in ft8_sync_score ft8_extract_likelihood ftx_normalize_logl; do
    has "\\b$n[[:space:]]*\\(" "$UP/ft8/decode.c" || missing="$missing $n"
done
has '\bbp_decode[[:space:]]*\(' "$UP/ft8/ldpc.h" || missing="$missing bp_decode(ft8/ldpc.h)"
if [ -n "$missing" ] && [ -z "${PIN_EXTRA_CFLAGS:-}" ]; then
    INT=""
    echo "NOTE: this revision does not define:$missing"
    echo "      -> comparing the PUBLIC interface only.  To include sync scores / LLRs / BP convergence, name its equivalents:"
    echo "         PIN_EXTRA_CFLAGS='-DPIN_UPSTREAM_SYNC_SCORE=<fn> -DPIN_UPSTREAM_EXTRACT_LLR=<fn> -DPIN_UPSTREAM_NORMALIZE=<fn>' $0 $UP"
fi
echo "comparison depth: $([ -n "$INT" ] && echo 'public interface + internals (sync scores, LLR bits, BP convergence)' || echo 'public interface only')"
[ -n "${PIN_PLAN_ONLY:-}" ] && exit 0
make -s -C "$ROOT/oracle"
python3 "$P/make_inputs.py" "$W" ${PIN_LIGHT:+--light}
CFLAGS="-O2 -std=gnu17 -ffp-contract=off -fno-fast-math ${PIN_EXTRA_CFLAGS:-}"      # one IEEE operation per float operation on both sides
# (a) upstream: its own sources; with internals decode.c comes in through upstream_internals.c (which wraps its file-local functions)
UPSRC=""
for f in constants crc ldpc unpack text pack encode; do [ -f "$UP/ft8/$f.c" ] && UPSRC="$UPSRC $UP/ft8/$f.c"; done
if [ -n "$INT" ] && ! gcc $CFLAGS $INT -I"$UP" -I"$P" "$P/pin_harness.c" "$P/upstream_internals.c" $UPSRC -lm -o "$W/pin_upstream" 2> "$W/internals_build.log"; then
    head -20 "$W/internals_build.log"
    echo "NOTE: the internals backend does not compile against this revision (log: $W/internals_build.log) -> public interface only"
    INT=""
fi
[ -n "$INT" ] || gcc $CFLAGS -I"$UP" -I"$P" "$P/pin_harness.c" "$UP/ft8/decode.c" $UPSRC -lm -o "$W/pin_upstream"
# (b) the oracle under the same names, at the same depth
gcc $CFLAGS $INT -I"$ROOT/include/ft8_lib" -I"$ROOT/oracle" -I"$P" "$P/pin_harness.c" "$P/oracle_as_ft8_lib.c" \
    -L"$ROOT/oracle" -lft8oracle -Wl,-rpath,"$ROOT/oracle" -lm -o "$W/pin_oracle"
"$W/pin_upstream" "$W/waterfalls.bin" "$W/messages.txt" > "$W/upstream.txt"
"$W/pin_oracle" "$W/waterfalls.bin" "$W/messages.txt" > "$W/oracle.txt"
echo "upstream revision: $(git -C "$UP" rev-parse HEAD 2>/dev/null || echo unknown)"
echo "records: $(wc -l < "$W/oracle.txt") (oracle), $(wc -l < "$W/upstream.txt") (upstream); dumps in $W"
if diff -u "$W/upstream.txt" "$W/oracle.txt" > "$W/pin.diff"; then
    echo "PINNED: the oracle and upstream ft8_lib agree on every record ($([ -n "$INT" ] && echo 'public interface + internals' || echo 'PUBLIC INTERFACE ONLY: see the note above'))"
else
    head -60 "$W/pin.diff"
    echo "DIFFERENT: $(grep -c '^[-+][^-+]' "$W/pin.diff") lines differ (full diff: $W/pin.diff)"
    exit 1
fi
