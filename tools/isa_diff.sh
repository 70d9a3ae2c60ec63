#!/bin/bash
# Is one kernel instantiation byte-for-byte what it was at another commit?  (development aid)
#   tools/isa_diff.sh <git-rev> [mangled-kernel-name-fragment [fragment-in-the-current-build]]     default: the fp32 4-waves-per-object PnP kernel (the bench's kernel)
# Used to land changes that are meant for OTHER instantiations only (16-bit storage, other wave counts) without touching the
# kernel the headline is measured on: the register allocation of this kernel reacts to almost anything.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REV=$1; FRAG=${2:-pnp_uncert_kernelIfLi4ELb0E}; NEWFRAG=${3:-$FRAG}      # third argument: the fragment in the CURRENT build when the mangled name changed
T=$(mktemp -d)
mkdir -p $T/old
for f in monorun_pnp.hip pnp_kernel.inc pnp_kernel_body.inc pnp6_kernel.inc pnp_noc_kernel.inc kitti_eval_kernel.inc hessian_kernel.inc epnp_kernel.inc epnp_eig_lanes.inc epnp_eig_low4.inc epnp_stages.inc; do
    git -C $ROOT show $REV:monorun_amd/csrc/$f > $T/old/$f 2>/dev/null || true
done
git -C $ROOT show $REV:include/monorun_pnp.h > $T/old/monorun_pnp.h          # the old sources against the old header (prototypes change)
asm() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I $3 -S --cuda-device-only $1 -o $2 2>/dev/null; }
asm $T/old/monorun_pnp.hip $T/old.s $T/old
asm $ROOT/monorun_amd/csrc/monorun_pnp.hip $T/new.s $ROOT/include
 body() { awk -v frag="$2" '$0 ~ "^_Z.*" frag ".*:" {on=1} on {print} on && /s_endpgm/ {exit}' $1 | grep -v '^\s*;\|^\.L\|; %bb' | sed 's/;.*//; s/\.LBB[0-9]*_/.LBB_/g'; }      # branch-target labels carry the kernel's ordinal in the file: normalised
body $T/old.s "$FRAG" > $T/old.k; body $T/new.s "$NEWFRAG" > $T/new.k
echo "$(wc -l < $T/old.k) instructions at $REV, $(wc -l < $T/new.k) now, $(diff $T/old.k $T/new.k | grep -c '^[<>]') differing lines"
rm -rf $T
