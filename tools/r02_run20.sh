set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02u; mkdir -p $O
for T in 10752 10240 9216 8192; do timeout 60 ./build/local_sort_proto 28 $T 128 >> $O/local_proto.txt 2>&1; done
timeout 60 ./build/local_sort_proto_nt 28 10752 128 >> $O/local_proto.txt 2>&1
timeout 60 ./build/local_sort_proto 27 12288 64 >> $O/local_proto.txt 2>&1
timeout 60 ./build/local_sort_proto 26 12288 64 >> $O/local_proto.txt 2>&1
timeout 60 ./build/local_sort_proto 24 12288 16 >> $O/local_proto.txt 2>&1
timeout 60 ./build/local_sort_proto 28 11264 128 >> $O/local_proto.txt 2>&1
cat $O/local_proto.txt
