//! `spiral_rs::server` over libspiral_hip.so (MI355X).  See hip.rs.
pub mod hip;
