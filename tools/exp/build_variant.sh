# build a variant of the library beside the product one:  bash tools/exp/build_variant.sh NAME "XFLAGS" [STAMP]
# -> pointnet12_amd/libpn2_hip_NAME.so (git-ignored; use with PN2_LIB_PATH)
set -e
NAME=$1; XF=$2; ST=$3
D=/tmp/pn2_build_$NAME
rm -rf $D; mkdir -p $D/pointnet12_amd $D/include
cp -r pointnet12_amd/csrc $D/pointnet12_amd/csrc
cp include/*.h $D/include/
rm -f $D/pointnet12_amd/csrc/*.o
make -C $D/pointnet12_amd/csrc -j6 XFLAGS="$XF" STAMP=$ST > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
cp $D/pointnet12_amd/libpn2_hip.so pointnet12_amd/libpn2_hip_$NAME.so
echo built pointnet12_amd/libpn2_hip_$NAME.so
