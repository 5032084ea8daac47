// oracle/ref_prelude.hpp -- the ONLY hand-written code inside oracle/_ref (test infrastructure).
// Everything else in libk4ref.so is the reference's own engine source, respelled by oracle/make_ref.py.
// Each member below stands in for a reference member whose body is a call into the .NET runtime
// (list = make_ref.py EXCLUDED; paths relative to /root/reference/src/K4os.Compression.LZ4/).
#pragma once
#include <alloca.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace k4ref {
// C# built-in integer names (ECMA-334 8.3): fixed widths, `long` is 64-bit as in LP64
typedef uint8_t byte;
typedef int8_t sbyte;
typedef uint16_t ushort;
typedef uint32_t uint;
typedef uint64_t ulong;

#ifndef K4REF_ASSERT
#define K4REF_ASSERT(v) ((void) (v))          // [Conditional("DEBUG")]: a release build drops the call
#endif

// C# lets an enum be compared with the literal 0 (ECMA-334 10.2.4); `enum class` needs it spelled out
template <class E> constexpr typename std::enable_if<std::is_enum<E>::value, bool>::type operator!=(E a, int b) { return (int) a != b; }
template <class E> constexpr typename std::enable_if<std::is_enum<E>::value, bool>::type operator==(E a, int b) { return (int) a == b; }

// R16: try { A } finally { B }
template <class F> struct K4RefFinally { F f; K4RefFinally(F g) : f(g) {} ~K4RefFinally() { f(); } };

// Internal/Mem.cs:47-52,:69-71,:79-81,:103-106,:112-114,:144-145,:151-162
#define K4REF_MEMBERS_Mem \
	static constexpr bool System32 = sizeof(void*) < sizeof(ulong);                                          /* :50-52 */ \
	static void CpBlk(void* target, const void* source, uint length) { memcpy(target, source, length); }   /* :69-71 Unsafe.CopyBlockUnaligned */ \
	static void ZBlk(void* target, byte value, uint length) { memset(target, value, length); }             /* :79-81 Unsafe.InitBlockUnaligned */ \
	static void Move(byte* target, byte* source, int length) { memmove(target, source, (size_t) length); } /* :103-106 Buffer.MemoryCopy */ \
	static void* Alloc(int size) { return malloc((size_t) size); }                                         /* :112-114 Marshal.AllocHGlobal */ \
	static void Free(void* ptr) { free(ptr); }                                                             /* :144-145 Marshal.FreeHGlobal */ \
	template <class T, size_t N> static T* CloneArray(T (&array)[N]) {                                     /* :151-162 */ \
		T* target = (T*) Alloc((int) sizeof(array)); memcpy(target, array, sizeof(array)); return target; }

// Engine/LL.tools.cs:21-27 (Assert; Enforce32 / Algorithm, :29-36, are the managed dispatch the C entry points replace)
#define K4REF_MEMBERS_LL \
	static void Assert(bool value) { K4REF_ASSERT(value); }

// Internal/PinnedMemory.cs:63-77 (Alloc), :31-33 (Reference<T>), :121-134 (Free): pooled or native, not zeroed when zero == false
struct PinnedMemory {
	byte* _pointer = nullptr;
	static void Alloc(PinnedMemory& memory, int size, bool zero = true) { memory._pointer = (byte*) (zero ? calloc(1, (size_t) size) : malloc((size_t) size)); }
	template <class T> T* Reference() { return (T*) _pointer; }
	void Free() { free(_pointer); _pointer = nullptr; }
};

// System.Numerics.BitOperations.TrailingZeroCount(ulong) (used by LL64.tools.cs:57-59 under NET5_0_OR_GREATER)
struct BitOperations {
	static int TrailingZeroCount(ulong v) { return v ? __builtin_ctzll(v) : 64; }
};
} // namespace k4ref
