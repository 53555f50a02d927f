/*
 * s2c_scene.h -- C ABI of the on-device training-item assembly in libs2c_hip.so
 * (scan2cap_amd/csrc/s2c_scene.hip).
 *
 * Replaces, for a whole batch at once and from scenes RESIDENT in HBM, what the
 * reference's `ScannetReferenceDataset.__getitem__` does per item on a DataLoader
 * worker in numpy (lib/dataset.py:320-540) followed by a 173 MB/step host-to-device
 * copy (lib/solver.py:280-287).  The random draws stay on the host (numpy, in the
 * reference's order: utils/pc_utils.py:36-37, lib/dataset.py:398-424, :273-275) and
 * are handed over as plain arrays, so an item is a pure function of (scene, draws).
 *
 * Scene store layout (device memory, all scenes concatenated):
 *   verts    (V, vert_cols) f32   xyz rgb [normal]      `_aligned_vert.npy`
 *   mv       (V, Cm) f32          multiview features    enet_feats_maxpool.hdf5 rows
 *   ins, sem (V) i32              instance / nyu40 ids  `_ins_label.npy`, `_sem_label.npy`
 *   boxes    (NB, 8) f64          cx cy cz dx dy dz nyu40id object_id  `_aligned_bbox.npy`
 *   vert_off (S+1) i64, box_off (S+1) i32               prefix offsets per scene
 *   floor    (S) f32              np.percentile(z, 0.99) of each scene (s2c_scene_floor_height)
 *
 * aug (B, 32) f64 per item: [0] flip_x, [1] flip_y (0/1), [2..10] rotx, [11..19] roty,
 * [20..28] rotz (3x3 row-major), [29..31] translation.
 *
 * Conventions as in s2c_ops.h: device pointers, asynchronous on `stream` (hipStream_t
 * as void*), 0 on success / non-zero on a bad argument or a failed launch.
 */
#ifndef S2C_SCENE_H
#define S2C_SCENE_H
#ifdef __cplusplus
extern "C" {
#endif

#define S2C_SCENE_MAX_NUM_OBJ 128   /* lib/dataset.py:27 */
#define S2C_SCENE_MAX_INSTANCE 2048 /* instance ids must be < this */

/* floor[0] = np.percentile(verts[:, 2], 0.99) for one scene of nv vertices
 * (lib/dataset.py:360): exact order statistics by radix select, interpolated with
 * numpy's float32 arithmetic.  Run once per scene when it is registered. */
int s2c_scene_floor_height(long long nv, const float *verts, int vert_cols, float *floor,
                           void *stream);

/* choices (B, N) i64: N vertex indices of scene scene_ids[b] drawn on the device with the
 * distribution of utils/pc_utils.py:32-40 (distinct, random order; with replacement only
 * when the scene has fewer than N vertices) from seeds (B) u64 -- a keyed Feistel
 * permutation of [0, Nv) evaluated at 0..N-1, not numpy's stream.  Scenes < 2^31 vertices. */
int s2c_scene_sample(int B, int N, const long long *vert_off, const int *scene_ids,
                     const unsigned long long *seeds, long long *choices, void *stream);

/* cloud (B, N, Cout) f32, Cout = 3 + 3*use_color + 3*use_normal + Cm*use_multiview +
 * use_height: row (b, i) = vertex choices[b, i] of scene scene_ids[b] with
 *   xyz      flipped / rotated about x, y, z / translated (each rotation in f64, rounded
 *            to f32, as numpy does; lib/dataset.py:396-426) when `augment`,
 *   rgb      (rgb - MEAN_COLOR_RGB) / 256                    (lib/dataset.py:342-344),
 *   normal, multiview copied                                  (lib/dataset.py:346-357),
 *   height   z_raw - floor[scene]                             (lib/dataset.py:359-362). */
int s2c_scene_gather(int B, int N, int vert_cols, int Cm, int use_color, int use_normal,
                     int use_multiview, int use_height, int augment, const float *verts,
                     const float *mv, const long long *vert_off, const float *floor,
                     const int *scene_ids, const long long *choices, const double *aug,
                     float *cloud, void *stream);

/* Votes (lib/dataset.py:434-443): for every instance id among the sampled points whose
 * FIRST sampled point has a nyu40 id in `vote_id_mask` (bit i = id i votes), the vote
 * of each of its points is 0.5*(min+max) of the instance's sampled (augmented) xyz minus
 * the point.  vote_label (B,N,9) f32 = the vote three times, vote_label_mask (B,N) i64.
 * cloud row stride = Cout floats.  workspace: s2c_scene_votes_workspace_bytes(B) bytes of
 * device memory (per-instance tables; initialised by the callee). */
long long s2c_scene_votes_workspace_bytes(int B);
int s2c_scene_votes(int B, int N, int Cout, const float *cloud, const int *ins,
                    const int *sem, const long long *vert_off, const int *scene_ids,
                    const long long *choices, unsigned long long vote_id_mask,
                    void *workspace, float *vote_label, long long *vote_label_mask,
                    void *stream);

/* per-box labels; every pointer is (B, 128, ...) unless noted */
typedef struct {
  float *center_label;                    /* (B,128,3) */
  long long *size_class_label;            /* (B,128) */
  float *size_residual_label;             /* (B,128,3) */
  long long *sem_cls_label;               /* (B,128) */
  long long *scene_object_ids;            /* (B,128) */
  float *scene_object_rotations;          /* (B,128,9) */
  long long *scene_object_rotation_masks; /* (B,128) */
  float *box_label_mask;                  /* (B,128) */
  long long *ref_box_label;               /* (B,128) */
  double *gt_box_corner_label;            /* (B,128,8,3) */
  long long *gt_box_masks;                /* (B,128) */
  long long *gt_box_object_ids;           /* (B,128) */
  long long *num_bbox;                    /* (B) */
  float *ref_center_label;                /* (B,3) */
  long long *ref_size_class_label;        /* (B) */
  float *ref_size_residual_label;         /* (B,3) */
  double *ref_box_corner_label;           /* (B,8,3) */
} s2c_scene_labels;

/* Box labels of a batch (lib/dataset.py:371-393, :396-426 on the boxes via
 * model_util_scannet.py:47-79, :445-503): all arithmetic in f64 with numpy's
 * accumulation order.  box_rot (NB,9) f32 / box_rot_mask (NB) u8 = Scan2CAD rotations per
 * stored box (may be NULL), class_of_nyu40 (41) i32 (-1 = not a box class), mean_size
 * (num_class,3) f64, object_ids (B) i64 = the described object of each item.
 * Scenes hold 1..128 boxes with nyu40 ids of the table (the host mirror rejects anything
 * else when a scene is registered; the reference fails on such scenes too). */
int s2c_scene_box_labels(int B, int augment, const double *boxes, const int *box_off,
                         const float *box_rot, const unsigned char *box_rot_mask,
                         const int *scene_ids, const long long *object_ids,
                         const double *aug, const int *class_of_nyu40,
                         const double *mean_size, s2c_scene_labels out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
