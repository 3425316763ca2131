/* vltf_host.h -- C-ABI of libvltf_host.so: native TFRecord reading for the feeder.
 *
 * Replaces, for the hot path's input side, tf.python_io.tf_record_iterator (TF C++; dataset_.py:764) plus the
 * per-frame Python loop Dataset.deserialize_from_tfrecord / deserialize_example (dataset_.py:100-133,171-217):
 * one call reads a whole batch of frame records, verifies both masked CRC-32C checksums, decodes the
 * tf.train.Example fields the reference wrote (serialize.py:246-256) and copies the raw image bytes into a
 * caller buffer that is then uploaded and pre-processed on the device (vl_input_prep_u8).
 * Host pointers only; thread-safe (no global state besides a thread-local error string). */
#ifndef VLTF_HOST_H
#define VLTF_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* vlh_last_error(void);
/* CRC-32C (Castagnoli) and TFRecord's masked form ((c >> 15 | c << 17) + 0xa282ead8). */
uint32_t vlh_crc32c(const void* data, size_t n);
uint32_t vlh_masked_crc32c(const void* data, size_t n);
/* Reads `count` consecutive records of `path` starting at byte `offset`.
 *   images      : count * image_bytes bytes; record i's 'image_raw' must be exactly image_bytes long
 *   dims        : int32[count*3] <- height, width, depth
 *   labels      : int64[count*max_labels] <- 'label' values (first max_labels), label_counts[i] <- how many it had
 * Returns the byte offset after the last record read, or a negative code:
 *   -1 end of file before `count` records (records_read <- how many were complete), -2 I/O error,
 *   -3 CRC mismatch, -4 malformed Example / size mismatch.  records_read may be NULL. */
int64_t vlh_read_frames(const char* path, int64_t offset, int count, int verify_crc, uint8_t* images, int64_t image_bytes,
                        int32_t* dims, int64_t* labels, int max_labels, int32_t* label_counts, int32_t* records_read);
/* The same on `threads` threads (<= 64): the length headers are walked serially, the payload reads, checksums, Example parsing
 * and image copies are split over the threads.  Same results, return value and error codes; threads <= 1 = vlh_read_frames. */
int64_t vlh_read_frames_mt(const char* path, int64_t offset, int count, int verify_crc, uint8_t* images, int64_t image_bytes,
                           int32_t* dims, int64_t* labels, int max_labels, int32_t* label_counts, int32_t* records_read,
                           int threads);
/* Skips `count` records by their headers only (resume fast-forward, dataset_.py:772-811); returns the new offset
 * or a negative code as above. */
int64_t vlh_skip_records(const char* path, int64_t offset, int64_t count, int verify_crc);

#ifdef __cplusplus
}
#endif
#endif /* VLTF_HOST_H */
