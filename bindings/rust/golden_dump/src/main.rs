//! fl_golden_dump -- runs the REFERENCE crate (spiraldb/fastlanes, by path) on the exact inputs of
//! tests/golden/make_golden.py (`case_inputs`: splitmix64 streams of tests/datagen.py, 3 blocks per case) and prints the
//! SHA-256 of every output, one line per digest:
//!
//!     case <ty>/<w> <op> <sha256 of the little-endian output bytes>
//!     case <ty>/misc <delta|undelta|transpose|untranspose> <sha256>
//!     single <ty>_w<w> <sha256>          (unpack_single on every index of two blocks)
//!
//! tests/test_reference_crate_pins_golden.py compares these lines with tests/golden/golden.json: when they agree, the
//! committed golden vectors -- and through them the oracle and every GPU parity test -- are pinned by bytes the reference
//! itself produced (SURVEY.md 8(c) "residual risk").  SOURCE ONLY: the build image has no Rust toolchain, so this file has
//! never been compiled; it is run automatically the day `cargo` is on PATH.
#![allow(incomplete_features)]
#![feature(generic_const_exprs)]

use fastlanes::{BitPacking, Delta, FastLanes, FoR, Transpose};
use seq_macro::seq;

const N_BLOCKS: usize = 3;

// ---- tests/datagen.py: value k of stream `seed` = splitmix64 output k+1 of the generator seeded seed * GOLDEN ----------
const GOLDEN: u64 = 0x9E37_79B9_7F4A_7C15;
fn splitmix64(k: u64, seed: u64) -> u64 {
    let mut z = seed.wrapping_mul(GOLDEN).wrapping_add((k + 1).wrapping_mul(GOLDEN));
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58_476D_1CE4_E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D0_49BB_1331_11EB);
    z ^ (z >> 31)
}

trait Elem: Copy + Default + FastLanes + BitPacking + FoR + Delta + Transpose {
    const NAME: &'static str;
    fn from_u64(z: u64) -> Self; // truncating cast, like numpy's astype
    fn le_bytes(self, out: &mut Vec<u8>);
}
macro_rules! impl_elem {
    ($T:ty, $name:expr) => {
        impl Elem for $T {
            const NAME: &'static str = $name;
            fn from_u64(z: u64) -> Self { z as $T }
            fn le_bytes(self, out: &mut Vec<u8>) { out.extend_from_slice(&self.to_le_bytes()); }
        }
    };
}
impl_elem!(u8, "u8");
impl_elem!(u16, "u16");
impl_elem!(u32, "u32");
impl_elem!(u64, "u64");

fn values<T: Elem>(n: usize, seed: u64) -> Vec<T> {
    (0..n as u64).map(|k| T::from_u64(splitmix64(k, seed))).collect()
}

// ---- SHA-256 (FIPS 180-4), kept local so that the only dependencies are the reference and seq-macro --------------------
fn sha256(data: &[u8]) -> String {
    const K: [u32; 64] = [
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2,
    ];
    let mut h: [u32; 8] = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19];
    let mut msg = data.to_vec();
    let bits = (data.len() as u64).wrapping_mul(8);
    msg.push(0x80);
    while msg.len() % 64 != 56 {
        msg.push(0);
    }
    msg.extend_from_slice(&bits.to_be_bytes());
    for chunk in msg.chunks(64) {
        let mut w = [0u32; 64];
        for i in 0..16 {
            w[i] = u32::from_be_bytes([chunk[4 * i], chunk[4 * i + 1], chunk[4 * i + 2], chunk[4 * i + 3]]);
        }
        for i in 16..64 {
            let s0 = w[i - 15].rotate_right(7) ^ w[i - 15].rotate_right(18) ^ (w[i - 15] >> 3);
            let s1 = w[i - 2].rotate_right(17) ^ w[i - 2].rotate_right(19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16].wrapping_add(s0).wrapping_add(w[i - 7]).wrapping_add(s1);
        }
        let mut v = h;
        for i in 0..64 {
            let s1 = v[4].rotate_right(6) ^ v[4].rotate_right(11) ^ v[4].rotate_right(25);
            let ch = (v[4] & v[5]) ^ (!v[4] & v[6]);
            let t1 = v[7].wrapping_add(s1).wrapping_add(ch).wrapping_add(K[i]).wrapping_add(w[i]);
            let s0 = v[0].rotate_right(2) ^ v[0].rotate_right(13) ^ v[0].rotate_right(22);
            let maj = (v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]);
            let t2 = s0.wrapping_add(maj);
            v = [t1.wrapping_add(t2), v[0], v[1], v[2], v[3].wrapping_add(t1), v[4], v[5], v[6]];
        }
        for i in 0..8 {
            h[i] = h[i].wrapping_add(v[i]);
        }
    }
    h.iter().map(|x| format!("{x:08x}")).collect()
}

fn digest<T: Elem>(v: &[T]) -> String {
    let mut bytes = Vec::with_capacity(v.len() * core::mem::size_of::<T>());
    for x in v {
        x.le_bytes(&mut bytes);
    }
    sha256(&bytes)
}

/// Everything tests/golden/make_golden.py records for one (T, W): the reference's const-generic methods, block by block.
fn dump_width<T: Elem, const W: usize>()
where
    fastlanes::BitPackWidth<W>: fastlanes::SupportedBitPackWidth<T>,
    [(); 1024 * W / T::T]:,
    [(); T::LANES]:,
{
    let bits = T::T as u64;
    let seed = 1000 * bits + W as u64;
    let pl = 1024 * W / T::T;
    let vals: Vec<T> = values(N_BLOCKS * 1024, seed);
    let packed: Vec<T> = values(N_BLOCKS * pl, seed + 100_000);
    let refs: Vec<T> = values(N_BLOCKS, seed + 200_000);
    let bases: Vec<T> = values(N_BLOCKS * T::LANES, seed + 300_000);
    let (mut o_pack, mut o_for) = (vec![T::default(); N_BLOCKS * pl], vec![T::default(); N_BLOCKS * pl]);
    let (mut o_unpack, mut o_unfor, mut o_undelta) =
        (vec![T::default(); N_BLOCKS * 1024], vec![T::default(); N_BLOCKS * 1024], vec![T::default(); N_BLOCKS * 1024]);
    for b in 0..N_BLOCKS {
        let v: &[T; 1024] = vals[b * 1024..(b + 1) * 1024].try_into().unwrap();
        let p: &[T; 1024 * W / T::T] = packed[b * pl..(b + 1) * pl].try_into().unwrap();
        let base: &[T; T::LANES] = bases[b * T::LANES..(b + 1) * T::LANES].try_into().unwrap();
        T::pack::<W>(v, (&mut o_pack[b * pl..(b + 1) * pl]).try_into().unwrap());
        T::for_pack::<W>(v, refs[b], (&mut o_for[b * pl..(b + 1) * pl]).try_into().unwrap());
        T::unpack::<W>(p, (&mut o_unpack[b * 1024..(b + 1) * 1024]).try_into().unwrap());
        T::unfor_pack::<W>(p, refs[b], (&mut o_unfor[b * 1024..(b + 1) * 1024]).try_into().unwrap());
        T::undelta_pack::<W>(p, base, (&mut o_undelta[b * 1024..(b + 1) * 1024]).try_into().unwrap());
    }
    let n = T::NAME;
    println!("case {n}/{W} pack {}", digest(&o_pack));
    println!("case {n}/{W} unpack {}", digest(&o_unpack));
    println!("case {n}/{W} for_pack {}", digest(&o_for));
    println!("case {n}/{W} unfor_pack {}", digest(&o_unfor));
    println!("case {n}/{W} undelta_pack {}", digest(&o_undelta));
    // unpack_single on every index of two blocks (make_golden.py "unpack_single": seed 3300 + 64*T + W)
    let pk2: Vec<T> = values(2 * pl, 3300 + 64 * bits + W as u64);
    let got: Vec<T> = (0..2048)
        .map(|i| {
            let p: &[T; 1024 * W / T::T] = pk2[(i / 1024) * pl..(i / 1024 + 1) * pl].try_into().unwrap();
            T::unpack_single::<W>(p, i % 1024)
        })
        .collect();
    println!("single {n}_w{W} {}", digest(&got));
}

fn dump_misc<T: Elem>()
where
    [(); T::LANES]:,
{
    let bits = T::T as u64;
    let seed = 1000 * bits + bits; // case_inputs(ty, T)
    let vals: Vec<T> = values(N_BLOCKS * 1024, seed);
    let bases: Vec<T> = values(N_BLOCKS * T::LANES, seed + 300_000);
    let mut outs = vec![vec![T::default(); N_BLOCKS * 1024]; 4];
    for b in 0..N_BLOCKS {
        let v: &[T; 1024] = vals[b * 1024..(b + 1) * 1024].try_into().unwrap();
        let base: &[T; T::LANES] = bases[b * T::LANES..(b + 1) * T::LANES].try_into().unwrap();
        T::delta(v, base, (&mut outs[0][b * 1024..(b + 1) * 1024]).try_into().unwrap());
        T::undelta(v, base, (&mut outs[1][b * 1024..(b + 1) * 1024]).try_into().unwrap());
        T::transpose(v, (&mut outs[2][b * 1024..(b + 1) * 1024]).try_into().unwrap());
        T::untranspose(v, (&mut outs[3][b * 1024..(b + 1) * 1024]).try_into().unwrap());
    }
    for (name, o) in ["delta", "undelta", "transpose", "untranspose"].iter().zip(outs.iter()) {
        println!("case {}/misc {name} {}", T::NAME, digest(o));
    }
}

fn main() {
    seq!(W in 0..=8 { dump_width::<u8, W>(); });
    seq!(W in 0..=16 { dump_width::<u16, W>(); });
    seq!(W in 0..=32 { dump_width::<u32, W>(); });
    seq!(W in 0..=64 { dump_width::<u64, W>(); });
    dump_misc::<u8>();
    dump_misc::<u16>();
    dump_misc::<u32>();
    dump_misc::<u64>();
}
