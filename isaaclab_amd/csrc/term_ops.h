/* Opcode tables of the lowered manager terms (shared by programs.hip and, through
 * isaaclab_amd/_terms.py which parses this file, by the Python term library).
 * Slot usage per opcode is documented next to each device implementation in programs.hip. */
#ifndef MDP_TERM_OPS_H
#define MDP_TERM_OPS_H

/* ---- observation ops: value of column k of the term --------------------------------------- */
#define MDP_OBS_COPY 0                     /* p0[e*i0 + i1 + idx(k)] */
#define MDP_OBS_SUB 1                      /* p0[e*i0 + i1 + idx(k)] - p1[e*i2 + i3 + idx(k)] */
#define MDP_OBS_BASE_LIN_VEL 2             /* LAB/envs/mdp/observations.py:40-44 */
#define MDP_OBS_BASE_ANG_VEL 3             /* :47-51 */
#define MDP_OBS_PROJECTED_GRAVITY 4        /* :54-58 */
#define MDP_OBS_ROOT_QUAT_UNIQUE 5         /* :68-82 (make_quat_unique=True) */
#define MDP_OBS_JOINT_POS_LIMIT_NORMALIZED 6 /* :124-137 */
#define MDP_OBS_HEIGHT_SCAN 7              /* :165-173, reads the sensor buffers */
#define MDP_OBS_HEIGHT_SCAN_CAST 8         /* :165-173 fused with ray_caster.py:220-260 */
#define MDP_OBS_ADD_ROW_BIAS 9            /* p0[e*i0 + i1 + idx(k)] + p1[e*i2 + i3]  (noise_model.py:173-182) */
#define MDP_OBS_SKIP 10                    /* columns written by another launch (height scan from the cast kernel) */
#define MDP_OBS_NUM_OPS 11

/* ---- reward ops: one scalar per env --------------------------------------------------------- */
#define MDP_REW_EXTERNAL 0
#define MDP_REW_IS_ALIVE 1                 /* LAB/envs/mdp/rewards.py:31-33 */
#define MDP_REW_IS_TERMINATED 2            /* :36-38 */
#define MDP_REW_IS_TERMINATED_TERM 3       /* :41-68 */
#define MDP_REW_LIN_VEL_Z_L2 4             /* :76-80 */
#define MDP_REW_ANG_VEL_XY_L2 5            /* :83-87 */
#define MDP_REW_FLAT_ORIENTATION_L2 6      /* :90-97 */
#define MDP_REW_BASE_HEIGHT_L2 7           /* :100-122 */
#define MDP_REW_BODY_LIN_ACC_L2 8          /* :125-128 */
#define MDP_REW_SUM_SQ 9                   /* :136-143,153-170,250-252 */
#define MDP_REW_SUM_ABS 10                 /* :146-150 */
#define MDP_REW_SUM_ABS_DIFF 11            /* :173-179,226-242 */
#define MDP_REW_SUM_SQ_DIFF 12             /* :245-247 */
#define MDP_REW_JOINT_POS_LIMITS 13        /* :182-196 */
#define MDP_REW_JOINT_VEL_LIMITS 14        /* :199-218 */
#define MDP_REW_UNDESIRED_CONTACTS 15      /* :260-268 */
#define MDP_REW_CONTACT_FORCES 16          /* :271-279 */
#define MDP_REW_TRACK_LIN_VEL_XY_EXP 17    /* :287-298 */
#define MDP_REW_TRACK_ANG_VEL_Z_EXP 18     /* :301-309 */
#define MDP_REW_FEET_AIR_TIME 19           /* TASKS/manager_based/locomotion/velocity/mdp/rewards.py:25-44 */
#define MDP_REW_FEET_AIR_TIME_POSITIVE_BIPED 20 /* :47-66 */
#define MDP_REW_FEET_SLIDE 21              /* :69-83 */
#define MDP_REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP 22 /* :86-96 */
#define MDP_REW_TRACK_ANG_VEL_Z_WORLD_EXP 23 /* :99-106 */
#define MDP_REW_NUM_OPS 24

/* ---- termination ops: one flag per env ------------------------------------------------------ */
#define MDP_TERM_EXTERNAL 0
#define MDP_TERM_TIME_OUT 1                /* LAB/envs/mdp/terminations.py:30-32 */
#define MDP_TERM_COMMAND_RESAMPLE 2        /* :35-42 */
#define MDP_TERM_BAD_ORIENTATION 3         /* :50-59 */
#define MDP_TERM_ROOT_HEIGHT_BELOW_MINIMUM 4 /* :62-72 */
#define MDP_TERM_JOINT_POS_OUT_OF_MANUAL_LIMIT 5 /* :90-105 */
#define MDP_TERM_JOINT_VEL_OUT_OF_LIMIT 6  /* :108-114 */
#define MDP_TERM_JOINT_VEL_OUT_OF_MANUAL_LIMIT 7 /* :117-124 */
#define MDP_TERM_JOINT_EFFORT_OUT_OF_LIMIT 8 /* :127-142 */
#define MDP_TERM_ILLEGAL_CONTACT 9         /* :150-158 */
#define MDP_TERM_TERRAIN_OUT_OF_BOUNDS 10  /* TASKS/manager_based/locomotion/velocity/mdp/terminations.py:24-52 */
#define MDP_TERM_NUM_OPS 11

#endif
