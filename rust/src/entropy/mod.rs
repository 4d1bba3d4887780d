pub mod ari;
