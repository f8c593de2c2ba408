#!/bin/bash
# Host sanitizers over the loader, the Partition build and the PNG writer (CPU only, g++):
#   bash exp/sanitize/run.sh [random-per-kind] [damaged]
#   SAN=thread bash exp/sanitize/run.sh ...     ThreadSanitizer instead (the tree build and the PNG writer use threads)
# Exit status 0 and no report = clean.  The device half (the host code of mirt_api.hip) is exp/sanitize/run_device.sh.
set -e
cd "$(dirname "$0")/../.."
T=$(mktemp -d /tmp/mirt_san.XXXXXX)
python3 exp/sanitize/make_texts.py "$T/texts" "${1:-40}" "${2:-200}"
SANFLAGS="-fsanitize=address,undefined -fno-sanitize-recover=undefined"
[ "$SAN" = thread ] && SANFLAGS="-fsanitize=thread"
g++ -O1 -g -std=c++14 -ffp-contract=off -pthread $SANFLAGS -fno-omit-frame-pointer \
    -o "$T/driver" exp/sanitize/driver.cpp raytracer_amd/csrc/mirt_host.cpp raytracer_amd/csrc/mirt_png.cpp -lz
ls "$T"/texts/* | xargs -n 200 "$T/driver"
rm -rf "$T"
