#!/bin/bash
# tools/mkvariant.sh NAME [git-rev]  -- build the csrc of a git revision (default: working tree) into build_ab/NAME.so for A/B timing
set -e
cd "$(dirname "$0")/.."
name=$1; rev=$2; extra=$3
tmp=build_ab/src_$name; rm -rf $tmp; mkdir -p $tmp/anatomask_amd/csrc $tmp/include
if [ -n "$rev" ]; then
  for f in $(git ls-tree --name-only $rev anatomask_amd/csrc/ include/); do git show $rev:$f > $tmp/$f; done
else
  cp anatomask_amd/csrc/* $tmp/anatomask_amd/csrc/; cp include/* $tmp/include/
fi
objs=""
for f in $tmp/anatomask_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics $extra -c $f -o ${f%.hip}.o 2>/dev/null & 
  objs="$objs ${f%.hip}.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/$name.so $objs
rm -rf $tmp
echo built build_ab/$name.so
