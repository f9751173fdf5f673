#!/bin/bash
# One-command pin of the oracle and of the GPU engine against the REAL kcftools (Java) on any machine that has a JDK 17+
# and Maven -- the build image has neither, which is why the parity of this repository is "unpinned" (DESIGN.md section 2).
#
#   KCFTOOLS_SRC=/path/to/kcftools-checkout  tools/pin_against_jar.sh [--no-gpu]      (builds the jar: needs Maven too)
#   KCFTOOLS_JAR=/path/to/kcftools-x.y.z.jar  tools/pin_against_jar.sh [--no-gpu]      (a prebuilt release jar: a JVM is enough)
#
# What it does, for every case of tests/golden/expected.json (3 databases x 3 window geometries, incl. -p and -c):
#   1. builds the reference jar (mvn -q -DskipTests package in $KCFTOOLS_SRC; pom.xml:20-55 pulls picocli & co from Maven Central)
#   2. runs   java -jar <jar> getVariations -r tests/golden/ref.fa -k tests/golden/<db> -o <tmp>/java_<case>.kcf -s S -f window -w W [-p P] -c C -t 2
#      (Plugins/GetVariants.java:21-61 is the option set)
#   3. compares every row of that KCF with tests/golden/expected.json (the oracle's integers, KD / SC to 2 decimals):
#      this is the check that has never been possible here -- it pins oracle/kcf_oracle.c, and through it every parity test
#   4. unless --no-gpu: runs kcftools_amd/bin/kcftools-amd getVariations with the same arguments and diffs the two KCF files
#      byte for byte, ignoring the ##date= and ##CMD= header lines (the only ones that may differ)
# Nothing from the reference is copied into this repository; the jar and the outputs live in a temporary directory.
# Exit status 0 = every case identical. The databases are this repository's synthetic ones (written by kcftools_amd/synth.py in
# the layout KMC.java:107-189 reads); to pin the reader against a database built by the real `kmc` tool, add KMC_DB=<prefix>
# REF_FA=<fasta>: the script then also runs both programs on that pair (window 10000) and diffs the output.
set -euo pipefail
HERE="$(cd "$(dirname "$0")/.." && pwd)"
GPU=1
[ "${1:-}" = "--no-gpu" ] && GPU=0
command -v java >/dev/null || { echo "pin: no java on PATH" >&2; exit 3; }
TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
if [ -n "${KCFTOOLS_JAR:-}" ]; then
    JAR="$KCFTOOLS_JAR"
    [ -f "$JAR" ] || { echo "pin: KCFTOOLS_JAR=$JAR is not a file" >&2; exit 3; }
else
    : "${KCFTOOLS_SRC:?set KCFTOOLS_JAR to a kcftools jar, or KCFTOOLS_SRC to a checkout of sivasubramanics/kcftools}"
    command -v mvn >/dev/null || { echo "pin: no mvn on PATH (or give a prebuilt jar: KCFTOOLS_JAR=...)" >&2; exit 3; }
    ( cd "$KCFTOOLS_SRC" && mvn -q -DskipTests package )
    JAR="$(ls "$KCFTOOLS_SRC"/target/*.jar | grep -v original | head -1)"
    [ -f "$JAR" ] || { echo "pin: no jar under $KCFTOOLS_SRC/target" >&2; exit 3; }
fi
echo "pin: reference jar $JAR ($(java -version 2>&1 | head -1))"
CLI="$HERE/kcftools_amd/bin/kcftools-amd"
G="$HERE/tests/golden"
status=0
strip() { grep -v -e '^##date=' -e '^##CMD=' "$1"; }
while IFS=$'\t' read -r idx db window step minc; do
    args=(-r "$G/ref.fa" -k "$G/$db" -s S -f window -w "$window" -c "$minc")
    [ "$step" != "0" ] && args+=(-p "$step")
    java -jar "$JAR" getVariations "${args[@]}" -o "$TMP/java_$idx.kcf" -t 2 > "$TMP/java_$idx.log" 2>&1 || { echo "pin: case $idx: the reference failed:"; tail -5 "$TMP/java_$idx.log"; status=1; continue; }
    if python3 "$HERE/tools/kcf_vs_expected.py" "$TMP/java_$idx.kcf" "$G/expected.json" "$idx"; then
        echo "pin: case $idx ($db -w $window -p $step -c $minc): reference == oracle expectations"
    else
        echo "pin: case $idx ($db -w $window -p $step -c $minc): reference DIFFERS from the oracle expectations"; status=1
    fi
    if [ "$GPU" = 1 ]; then
        "$CLI" getVariations "${args[@]}" -o "$TMP/amd_$idx.kcf" > "$TMP/amd_$idx.log" 2>&1 || { echo "pin: case $idx: kcftools-amd failed:"; tail -5 "$TMP/amd_$idx.log"; status=1; continue; }
        if diff <(strip "$TMP/java_$idx.kcf") <(strip "$TMP/amd_$idx.kcf") > "$TMP/diff_$idx.txt"; then
            echo "pin: case $idx: kcftools-amd KCF is byte-identical to the reference's (##date / ##CMD aside)"
        else
            echo "pin: case $idx: KCF files differ:"; head -10 "$TMP/diff_$idx.txt"; status=1
        fi
    fi
done < <(python3 "$HERE/tools/kcf_vs_expected.py" --list "$G/expected.json")
if [ -n "${KMC_DB:-}" ] && [ -n "${REF_FA:-}" ]; then
    java -jar "$JAR" getVariations -r "$REF_FA" -k "$KMC_DB" -s S -f window -w 10000 -o "$TMP/java_real.kcf" -t 8 > "$TMP/java_real.log" 2>&1
    "$CLI" getVariations -r "$REF_FA" -k "$KMC_DB" -s S -f window -w 10000 -o "$TMP/amd_real.kcf" > "$TMP/amd_real.log" 2>&1
    if diff <(strip "$TMP/java_real.kcf") <(strip "$TMP/amd_real.kcf") > "$TMP/diff_real.txt"; then
        echo "pin: real KMC database $KMC_DB: byte-identical KCF"
    else
        echo "pin: real KMC database $KMC_DB: KCF files differ:"; head -10 "$TMP/diff_real.txt"; status=1
    fi
fi
[ "$status" = 0 ] && echo "pin: ALL CASES IDENTICAL -- the oracle (and the engine) are pinned against $JAR" || echo "pin: FAILED"
exit $status
