cd $GRAFT_REPO_ROOT
for mib in 32 64 96 160 256 512; do
  SPZ_AMD_HOST_CHUNK_MIB=$mib ./spz_amd/bin/host_bench 10000000 3 4 > gpurun_out/host_bench_chunk_$mib.json 2>&1
  python3 -c "import json; d=json.load(open('gpurun_out/host_bench_chunk_$mib.json')); print($mib, 'MiB: abi enc', d['abi_encode_host_s'], 'dec', d['abi_decode_host_s'], 'pack fresh', d['pack_to_stream_fresh_vector_s'], 'reused', d['pack_to_stream_reused_vector_s'], 'unpack', d['unpack_from_stream_s'], 'convert', d['convert_coordinates_s'])"
done
