/*
 * o_kernels.c -- the reference's RNS kernels restated as plain loops
 * (grid dimensions made explicit, same index arithmetic).
 * TEST INFRASTRUCTURE ONLY (see hegpu_oracle.h).
 */
#include "hegpu_oracle.h"
#include "o_kernels.h"

/* addition.cu:10-21 ; grid (N/256, limbs, parts) */
void o_addition(const u64* a, const u64* b, u64* out, const omod_t* mods,
                int n_power, int limbs, int parts)
{
    u64 n = ((u64) 1) << n_power;
    for (int z = 0; z < parts; z++)
        for (int y = 0; y < limbs; y++)
            for (u64 x = 0; x < n; x++) {
                u64 loc = x + ((u64) y << n_power) +
                          (((u64) limbs * z) << n_power);
                out[loc] = o_add(a[loc], b[loc], &mods[y]);
            }
}

/* addition.cu:23-34 */
void o_substraction(const u64* a, const u64* b, u64* out, const omod_t* mods,
                    int n_power, int limbs, int parts)
{
    u64 n = ((u64) 1) << n_power;
    for (int z = 0; z < parts; z++)
        for (int y = 0; y < limbs; y++)
            for (u64 x = 0; x < n; x++) {
                u64 loc = x + ((u64) y << n_power) +
                          (((u64) limbs * z) << n_power);
                out[loc] = o_sub(a[loc], b[loc], &mods[y]);
            }
}

/* addition.cu:36-47 : sub(0, x) */
void o_negation(const u64* a, u64* out, const omod_t* mods, int n_power,
                int limbs, int parts)
{
    u64 n = ((u64) 1) << n_power;
    for (int z = 0; z < parts; z++)
        for (int y = 0; y < limbs; y++)
            for (u64 x = 0; x < n; x++) {
                u64 loc = x + ((u64) y << n_power) +
                          (((u64) limbs * z) << n_power);
                out[loc] = o_sub(0, a[loc], &mods[y]);
            }
}

/* multiplication.cu:102-126 ; grid (N/256, decomp_size, 1) */
void o_cross_multiplication(const u64* in1, const u64* in2, u64* out,
                            const omod_t* mods, int n_power, int decomp_size)
{
    u64 n = ((u64) 1) << n_power;
    u64 part = (u64) decomp_size << n_power;
    for (int y = 0; y < decomp_size; y++) {
        const omod_t* m = &mods[y];
        for (u64 x = 0; x < n; x++) {
            u64 loc = x + ((u64) y << n_power);
            u64 a0 = in1[loc], a1 = in1[loc + part];
            u64 b0 = in2[loc], b1 = in2[loc + part];
            u64 o0 = o_mult(a0, b0, m);
            u64 o10 = o_mult(a0, b1, m);
            u64 o11 = o_mult(a1, b0, m);
            u64 o2 = o_mult(a1, b1, m);
            out[loc] = o0;
            out[loc + part] = o_add(o10, o11, m);
            out[loc + 2 * part] = o2;
        }
    }
}

/* switchkey.cu:11-27 cipher_broadcast_kernel ; grid (N/256, Q, 1).
 * Reduces with mult(1, x, q_i). */
void o_cipher_broadcast(const u64* input, u64* output, const omod_t* mods,
                        int n_power, int Q, int rns_mod_count)
{
    u64 n = ((u64) 1) << n_power;
    for (int y = 0; y < Q; y++) {
        u64 location = ((u64) rns_mod_count * y) << n_power;
        for (u64 x = 0; x < n; x++) {
            u64 in = input[x + ((u64) y << n_power)];
            for (int i = 0; i < rns_mod_count; i++)
                output[x + ((u64) i << n_power) + location] =
                    o_mult(1, in, &mods[i]);
        }
    }
}

/* switchkey.cu:29-59 cipher_broadcast_leveled_kernel ; grid (N/256, l, 1) */
void o_cipher_broadcast_leveled(const u64* input, u64* output,
                                const omod_t* mods, int first_rns_mod_count,
                                int current_rns_mod_count, int n_power,
                                int grid_y)
{
    u64 n = ((u64) 1) << n_power;
    int level = first_rns_mod_count - current_rns_mod_count;
    for (int y = 0; y < grid_y; y++) {
        u64 location = ((u64) current_rns_mod_count * y) << n_power;
        for (u64 x = 0; x < n; x++) {
            u64 in = input[x + ((u64) y << n_power)];
            for (int i = 0; i < current_rns_mod_count; i++) {
                int mod_index = (i < grid_y) ? i : i + level;
                output[x + ((u64) i << n_power) + location] =
                    o_reduce_forced(in, &mods[mod_index]);
            }
        }
    }
}

/* switchkey.cu:61-162 keyswitch_multiply_accumulate_kernel ;
 * grid (N/256, Q', 1).  The 4x-unrolled loop accumulates digits in order
 * 0..Q-1 with canonical modular adds; restated as one loop. */
void o_keyswitch_mac(const u64* input, const u64* key, u64* output,
                     const omod_t* mods, int n_power, int Qt, int digits)
{
    u64 n = ((u64) 1) << n_power;
    u64 key_offset1 = (u64) Qt << n_power;
    u64 key_offset2 = (u64) Qt << (n_power + 1);
    for (int y = 0; y < Qt; y++) {
        const omod_t* m = &mods[y];
        for (u64 x = 0; x < n; x++) {
            u64 index = x + ((u64) y << n_power);
            u64 s0 = 0, s1 = 0;
            for (int i = 0; i < digits; i++) {
                u64 in = input[index + (((u64) i * Qt) << n_power)];
                u64 k0 = key[index + key_offset2 * i];
                u64 k1 = key[index + key_offset2 * i + key_offset1];
                s0 = o_add(s0, o_mult(in, k0, m), m);
                s1 = o_add(s1, o_mult(in, k1, m), m);
            }
            output[index] = s0;
            output[index + key_offset1] = s1;
        }
    }
}

/* switchkey.cu:164-285 keyswitch_multiply_accumulate_leveled_kernel ;
 * grid (N/256, l+1, 1); last row maps to the P limb of the key. */
void o_keyswitch_mac_leveled(const u64* input, const u64* key, u64* output,
                             const omod_t* mods, int first_rns_mod_count,
                             int current_decomp_mod_count, int n_power)
{
    u64 n = ((u64) 1) << n_power;
    int cur1 = current_decomp_mod_count + 1;
    u64 key_offset1 = (u64) first_rns_mod_count << n_power;
    u64 key_offset2 = (u64) first_rns_mod_count << (n_power + 1);
    for (int y = 0; y < cur1; y++) {
        int key_index = (y == current_decomp_mod_count)
                            ? (first_rns_mod_count - 1)
                            : y;
        const omod_t* m = &mods[key_index];
        for (u64 x = 0; x < n; x++) {
            u64 index = x + ((u64) y << n_power);
            u64 s0 = 0, s1 = 0;
            for (int i = 0; i < current_decomp_mod_count; i++) {
                u64 in = input[index + (((u64) i * cur1) << n_power)];
                u64 kb = x + ((u64) key_index << n_power) + key_offset2 * i;
                s0 = o_add(s0, o_mult(in, key[kb], m), m);
                s1 = o_add(s1, o_mult(in, key[kb + key_offset1], m), m);
            }
            output[index] = s0;
            output[index + ((u64) cur1 << n_power)] = s1;
        }
    }
}

/* switchkey.cu:400-478 divide_round_lastq_kernel (+ _switchkey variant:
 * add ct only to part 0) ; grid (N/256, Q, 2) */
void o_divide_round_lastq(const u64* input, const u64* ct, u64* output,
                          const omod_t* mods, const u64* half,
                          const u64* half_mod, const u64* last_q_modinv,
                          int n_power, int decomp_mod_count, int switchkey)
{
    u64 n = ((u64) 1) << n_power;
    int D = decomp_mod_count;
    for (int z = 0; z < 2; z++)
        for (int y = 0; y < D; y++)
            for (u64 x = 0; x < n; x++) {
                u64 last = input[x + ((u64) D << n_power) +
                                 (((u64) (D + 1) << n_power) * z)];
                last = o_add(last, half[0], &mods[D]);
                last = o_reduce_forced(last, &mods[y]);
                last = o_sub(last, half_mod[y], &mods[y]);
                u64 in = input[x + ((u64) y << n_power) +
                               (((u64) (D + 1) << n_power) * z)];
                in = o_sub(in, last, &mods[y]);
                in = o_mult(in, last_q_modinv[y], &mods[y]);
                u64 loc = x + ((u64) y << n_power) + (((u64) D << n_power) * z);
                u64 c = (switchkey && z != 0) ? 0 : ct[loc];
                output[loc] = o_add(c, in, &mods[y]);
            }
}

/* switchkey.cu:678-705 divide_round_lastq_leveled_stage_one_kernel ;
 * grid (N/256, 2, 1) */
void o_divide_round_lastq_leveled_stage_one(
    const u64* input, u64* output, const omod_t* mods, const u64* half,
    const u64* half_mod, int n_power, int first_decomp_count,
    int current_decomp_count)
{
    u64 n = ((u64) 1) << n_power;
    int C = current_decomp_count;
    for (int y = 0; y < 2; y++)
        for (u64 x = 0; x < n; x++) {
            u64 last = input[x + ((u64) C << n_power) +
                             (((u64) (C + 1) << n_power) * y)];
            last = o_add(last, half[0], &mods[first_decomp_count]);
            for (int i = 0; i < C; i++) {
                u64 li = o_reduce_forced(last, &mods[i]);
                li = o_sub(li, half_mod[i], &mods[i]);
                output[x + ((u64) i << n_power) +
                       (((u64) C << n_power) * y)] = li;
            }
        }
}

/* switchkey.cu:707-771 ..._stage_two_kernel (+ _switchkey) ;
 * grid (N/256, l, 2).  ct and output may alias (in-place relin). */
void o_divide_round_lastq_leveled_stage_two(
    const u64* input_last, const u64* input, const u64* ct, u64* output,
    const omod_t* mods, const u64* last_q_modinv, int n_power,
    int current_decomp_count, int switchkey)
{
    u64 n = ((u64) 1) << n_power;
    int C = current_decomp_count;
    for (int z = 0; z < 2; z++)
        for (int y = 0; y < C; y++)
            for (u64 x = 0; x < n; x++) {
                u64 loc = x + ((u64) y << n_power) + (((u64) C << n_power) * z);
                u64 last = input_last[loc];
                u64 in = input[x + ((u64) y << n_power) +
                               (((u64) (C + 1) << n_power) * z)];
                in = o_sub(in, last, &mods[y]);
                in = o_mult(in, last_q_modinv[y], &mods[y]);
                u64 c = (switchkey && z != 0) ? 0 : ct[loc];
                output[loc] = o_add(c, in, &mods[y]);
            }
}

/* switchkey.cu:776-790 move_cipher_leveled_kernel ; grid (N/256, C, 2) */
void o_move_cipher_leveled(const u64* input, u64* output, int n_power,
                           int current_decomp_count)
{
    u64 n = ((u64) 1) << n_power;
    int C = current_decomp_count;
    for (int z = 0; z < 2; z++)
        for (int y = 0; y < C; y++)
            for (u64 x = 0; x < n; x++) {
                u64 loc = x + ((u64) y << n_power) +
                          (((u64) (C + 1) << n_power) * z);
                output[loc] = input[loc];
            }
}

/* switchkey.cu:792-815 divide_round_lastq_rescale_kernel ;
 * grid (N/256, C, 2) */
void o_divide_round_lastq_rescale(const u64* input_last, const u64* input,
                                  u64* output, const omod_t* mods,
                                  const u64* last_q_modinv, int n_power,
                                  int current_decomp_count)
{
    u64 n = ((u64) 1) << n_power;
    int C = current_decomp_count;
    for (int z = 0; z < 2; z++)
        for (int y = 0; y < C; y++)
            for (u64 x = 0; x < n; x++) {
                u64 loc = x + ((u64) y << n_power) + (((u64) C << n_power) * z);
                u64 last = input_last[loc];
                u64 in = input[x + ((u64) y << n_power) +
                               (((u64) (C + 1) << n_power) * z)];
                in = o_sub(in, last, &mods[y]);
                output[loc] = o_mult(in, last_q_modinv[y], &mods[y]);
            }
}

/* switchkey.cu:1558-1590 ckks_duplicate_kernel ; grid (N/256, l, 1) */
void o_ckks_duplicate(const u64* cipher, u64* output, const omod_t* mods,
                      int n_power, int first_rns_mod_count,
                      int current_rns_mod_count, int current_decomp_mod_count)
{
    u64 n = ((u64) 1) << n_power;
    int level = first_rns_mod_count - current_rns_mod_count;
    for (int y = 0; y < current_decomp_mod_count; y++) {
        u64 location = ((u64) current_rns_mod_count * y) << n_power;
        for (u64 x = 0; x < n; x++) {
            u64 v = cipher[x + ((u64) y << n_power) +
                           ((u64) current_decomp_mod_count << n_power)];
            for (int i = 0; i < current_rns_mod_count; i++) {
                int mod_index = (i < current_decomp_mod_count) ? i : i + level;
                output[x + ((u64) i << n_power) + location] =
                    o_reduce_forced(v, &mods[mod_index]);
            }
        }
    }
}

/* switchkey.cu:1592-1619 bfv_duplicate_kernel ; grid (N/256, Q, 2) */
void o_bfv_duplicate(const u64* cipher, u64* output1, u64* output2,
                     const omod_t* mods, int n_power, int Q,
                     int rns_mod_count)
{
    u64 n = ((u64) 1) << n_power;
    for (int z = 0; z < 2; z++)
        for (int y = 0; y < Q; y++)
            for (u64 x = 0; x < n; x++) {
                u64 v = cipher[x + ((u64) y << n_power) +
                               (((u64) Q << n_power) * z)];
                if (z == 0) {
                    output1[x + ((u64) y << n_power)] = v;
                } else {
                    u64 location = ((u64) rns_mod_count * y) << n_power;
                    for (int i = 0; i < rns_mod_count; i++)
                        output2[x + ((u64) i << n_power) + location] =
                            o_reduce_forced(v, &mods[i]);
                }
            }
}

/* switchkey.cu:1621-1718 (ckks) / 1720-1813 (bfv)
 * divide_round_lastq_permute_*_kernel ; grid (N/256, Q_size, 2).
 * For bfv pass first_Q_prime_size = Q_prime_size, first_Q_size = Q_size.
 * `q - x` without a zero test is kept; idx*galois_elt is
 * 32-bit int arithmetic in the reference (quirk 5) -- only bits 0..n_power
 * are used, so unsigned wraparound gives the same bits. */
void o_divide_round_lastq_permute(const u64* input, const u64* input2,
                                  u64* output, const omod_t* mods,
                                  const u64* half, const u64* half_mod,
                                  const u64* last_q_modinv, int galois_elt,
                                  int n_power, int Q_prime_size, int Q_size,
                                  int first_Q_prime_size, int first_Q_size,
                                  int P_size)
{
    u64 n = ((u64) 1) << n_power;
    uint32_t mask = (uint32_t) (n - 1);
    for (int z = 0; z < 2; z++)
        for (int y = 0; y < Q_size; y++)
            for (u64 x = 0; x < n; x++) {
                u64 last_ct[15];
                for (int i = 0; i < P_size; i++)
                    last_ct[i] = input[x + ((u64) (Q_size + i) << n_power) +
                                       (((u64) Q_prime_size << n_power) * z)];
                u64 in = input[x + ((u64) y << n_power) +
                               (((u64) Q_prime_size << n_power) * z)];
                int location_ = 0;
                for (int i = 0; i < P_size; i++) {
                    u64 lh = last_ct[P_size - 1 - i];
                    lh = o_add(lh, half[i], &mods[first_Q_prime_size - 1 - i]);
                    for (int j = 0; j < (P_size - 1 - i); j++) {
                        const omod_t* mj = &mods[first_Q_size + j];
                        u64 t1 = o_reduce_forced(lh, mj);
                        t1 = o_sub(t1, half_mod[location_ + first_Q_size + j],
                                   mj);
                        t1 = o_sub(last_ct[j], t1, mj);
                        last_ct[j] = o_mult(
                            t1, last_q_modinv[location_ + first_Q_size + j],
                            mj);
                    }
                    u64 t1 = o_reduce_forced(lh, &mods[y]);
                    t1 = o_sub(t1, half_mod[location_ + y], &mods[y]);
                    t1 = o_sub(in, t1, &mods[y]);
                    in = o_mult(t1, last_q_modinv[location_ + y], &mods[y]);
                    location_ += (first_Q_prime_size - 1 - i);
                }
                u64 val = in;
                if (z == 0)
                    val = o_add(input2[x + ((u64) y << n_power)], in, &mods[y]);
                uint32_t index_raw = (uint32_t) x * (uint32_t) galois_elt;
                uint32_t index = index_raw & mask;
                if ((index_raw >> n_power) & 1) val = mods[y].value - val;
                output[index + ((u64) y << n_power) +
                       (((u64) Q_size << n_power) * z)] = val;
            }
}

/* multiplication.cu:10-100 fast_convertion ; grid (N/256, 4, 1) */
void o_fast_convertion(const octx_t* c, const u64* in1, const u64* in2,
                       u64* out1)
{
    int n_power = c->n_power;
    u64 n = c->n;
    int ib = c->Q_size, ob = c->bsk_size;
    const omod_t* ibase = c->mod;
    const omod_t* obase = c->bsk;
    const omod_t* mt = &c->m_tilde;
#pragma omp parallel for collapse(2) schedule(static)
    for (int idy = 0; idy < 4; idy++)
        for (u64 x = 0; x < n; x++) {
            u64 location = x + ((u64) ((idy % 2) * ib) << n_power);
            const u64* input = ((idy >> 1) == 0) ? in1 : in2;
            u64 temp[O_MAX_BSK], temp_[O_MAX_BSK], temp2[O_MAX_BSK + 1];
            for (int i = 0; i < ib; i++) {
                temp_[i] = input[location + ((u64) i << n_power)];
                temp[i] = o_mult(temp_[i], mt->value, &ibase[i]);
                temp[i] = o_mult(temp[i], c->inv_punctured_prod_mod_base[i],
                                 &ibase[i]);
            }
            for (int i = 0; i < ob; i++) {
                temp2[i] = 0;
                for (int j = 0; j < ib; j++) {
                    u64 mu = o_mult(temp[j],
                                    c->base_change_matrix_Bsk[j + i * ib],
                                    &obase[i]);
                    temp2[i] = o_add(temp2[i], mu, &obase[i]);
                }
            }
            temp2[ob] = 0;
            for (int j = 0; j < ib; j++) {
                u64 ti = o_reduce_forced(temp[j], mt);
                u64 mu = o_mult(ti, c->base_change_matrix_m_tilde[j], mt);
                temp2[ob] = o_add(temp2[ob], mu, mt);
            }
            u64 m_tilde_div_2 = mt->value >> 1;
            u64 r = o_mult(temp2[ob], c->inv_prod_q_mod_m_tilde, mt);
            r = mt->value - r;
            for (int i = 0; i < ob; i++) {
                u64 t3 = r;
                if (t3 >= m_tilde_div_2) {
                    t3 = obase[i].value - mt->value;
                    t3 = o_add(t3, r, &obase[i]);
                }
                t3 = o_mult(t3, c->prod_q_mod_Bsk[i], &obase[i]);
                t3 = o_add(temp2[i], t3, &obase[i]);
                temp2[i] = o_mult(t3, c->inv_m_tilde_mod_Bsk[i], &obase[i]);
            }
            u64 location2 = x + ((u64) (idy * (ob + ib)) << n_power);
            for (int i = 0; i < ib; i++)
                out1[location2 + ((u64) i << n_power)] = temp_[i];
            for (int i = 0; i < ob; i++)
                out1[location2 + ((u64) (i + ib) << n_power)] = temp2[i];
        }
}

/* multiplication.cu:128-272 fast_floor ; grid (N/256, 3, 1) */
void o_fast_floor(const octx_t* c, const u64* in_baseq_Bsk, u64* out1)
{
    int n_power = c->n_power;
    u64 n = c->n;
    int ib = c->Q_size, ob = c->bsk_size;
    const omod_t* ibase = c->mod;
    const omod_t* obase = c->bsk;
    const omod_t* msk = &obase[ob - 1];
    u64 t = c->plain_mod.value;
#pragma omp parallel for collapse(2) schedule(static)
    for (int idy = 0; idy < 3; idy++)
        for (u64 x = 0; x < n; x++) {
            u64 location_q = x + ((u64) (idy * (ib + ob)) << n_power);
            u64 location_Bsk = location_q + ((u64) ib << n_power);
            u64 reg_q[O_MAX_BSK], reg_Bsk[O_MAX_BSK], temp[O_MAX_BSK];
            u64 temp3[O_MAX_BSK], temp4[O_MAX_BSK + 1];
            for (int i = 0; i < ib; i++) {
                reg_q[i] =
                    o_mult(in_baseq_Bsk[location_q + ((u64) i << n_power)], t,
                           &ibase[i]);
                reg_q[i] = o_mult(reg_q[i], c->inv_punctured_prod_mod_base[i],
                                  &ibase[i]);
            }
            for (int i = 0; i < ob; i++)
                reg_Bsk[i] =
                    o_mult(in_baseq_Bsk[location_Bsk + ((u64) i << n_power)],
                           t, &obase[i]);
            for (int i = 0; i < ob; i++) {
                temp[i] = 0;
                for (int j = 0; j < ib; j++) {
                    u64 mu = o_mult(reg_q[j],
                                    c->base_change_matrix_Bsk[j + i * ib],
                                    &obase[i]);
                    temp[i] = o_add(temp[i], mu, &obase[i]);
                }
            }
            for (int i = 0; i < ob; i++) {
                u64 t2 = o_sub(obase[i].value, temp[i], &obase[i]);
                t2 = o_add(t2, reg_Bsk[i], &obase[i]);
                reg_Bsk[i] = o_mult(t2, c->inv_prod_q_mod_Bsk[i], &obase[i]);
            }
            for (int i = 0; i < ob - 1; i++)
                temp3[i] = o_mult(reg_Bsk[i], c->inv_punctured_prod_mod_B[i],
                                  &obase[i]);
            for (int i = 0; i < ib; i++) {
                temp4[i] = 0;
                for (int j = 0; j < ob - 1; j++) {
                    u64 t3 = o_reduce_forced(temp3[j], &ibase[i]);
                    u64 mu = o_mult(
                        t3, c->base_change_matrix_q[j + i * (ob - 1)],
                        &ibase[i]);
                    mu = o_reduce_forced(mu, &ibase[i]);
                    temp4[i] = o_add(temp4[i], mu, &ibase[i]);
                }
            }
            temp4[ib] = 0;
            for (int j = 0; j < ob - 1; j++) {
                u64 mu = o_mult(temp3[j], c->base_change_matrix_msk[j], msk);
                temp4[ib] = o_add(temp4[ib], mu, msk);
            }
            u64 alpha_sk = o_sub(msk->value, reg_Bsk[ob - 1], msk);
            alpha_sk = o_add(alpha_sk, temp4[ib], msk);
            alpha_sk = o_mult(alpha_sk, c->inv_prod_B_mod_m_sk, msk);
            u64 m_sk_div_2 = msk->value >> 1;
            for (int i = 0; i < ib; i++) {
                u64 obase_ = o_reduce_forced(msk->value, &ibase[i]);
                u64 temp4_ = o_reduce_forced(temp4[i], &ibase[i]);
                u64 alpha_sk_ = o_reduce_forced(alpha_sk, &ibase[i]);
                if (alpha_sk > m_sk_div_2) {
                    u64 inner = o_sub(obase_, alpha_sk_, &ibase[i]);
                    inner = o_mult(inner, c->prod_B_mod_q[i], &ibase[i]);
                    temp4[i] = o_add(temp4_, inner, &ibase[i]);
                } else {
                    u64 inner =
                        o_sub(ibase[i].value, c->prod_B_mod_q[i], &ibase[i]);
                    inner = o_mult(inner, alpha_sk_, &ibase[i]);
                    temp4[i] = o_add(temp4_, inner, &ibase[i]);
                }
            }
            u64 location_out = x + ((u64) (idy * ib) << n_power);
            for (int i = 0; i < ib; i++)
                out1[location_out + ((u64) i << n_power)] = temp4[i];
        }
}
