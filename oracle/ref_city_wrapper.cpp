/*
 * ref_city_wrapper.cpp — extern "C" export of the REFERENCE's own frozen CityHash (contrib/cityhash102), compiled in place from
 * /root/reference (never copied) into oracle/_ref/libchref_city.so by oracle/Makefile.  TEST INFRASTRUCTURE ONLY: pins the product's
 * own CityHash128 (clickhouse_amd/csrc/feed.hip) -- the checksum of every compressed frame (CompressedReadBufferBase.cpp:49-51).
 */
#include <city.h>

extern "C" void ref_CityHash128(const char * data, unsigned long long size, unsigned long long out_low_high[2])
{
    const auto h = CityHash_v1_0_2::CityHash128(data, size);
    out_low_high[0] = h.low64;
    out_low_high[1] = h.high64;
}
